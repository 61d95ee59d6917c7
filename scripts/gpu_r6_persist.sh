#!/bin/bash
# Round 6: the persistent fused kernel (raftx_kp_f*) against the one-workgroup-per-pair launches, same box.
# 1. a small parity test first (a re-entry that went wrong must not take the whole visit with it), 2. the GPU suite,
# 3. kernel time against the pairs of a launch, both forms, 4. the default bench, both forms.
TAG=${1:-r06_persist}
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; cd $R
( timeout 180 python -m pytest tests/test_hip_parity.py -m gpu -x -q -k "synthetic_parity or live_reference" 2>&1 | tail -15 ) > $OUT/pytest_first.log
cat $OUT/pytest_first.log
grep -q "passed" $OUT/pytest_first.log || { echo "first parity run did not pass: stopping"; exit 1; }
( timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -25 ) > $OUT/pytest_gpu.log
cat $OUT/pytest_gpu.log
cat > /tmp/_row.py <<'PY'
import json, sys
d = json.loads(sys.stdin.read())
k = d['kernel_resident']
print(sys.argv[1], sys.argv[2], 'kernel_ms', round(k['kernel_ms'], 4), 'us_per_pair', round(1e3 * k['kernel_ms'] / int(sys.argv[2]), 4), 'mean_it', round(d['mean_iterations'], 3), flush=True)
PY
for n in 1250 2500 5000 10000 20000 40000; do
  for p in 0 1; do
    RAFTX_PERSIST=$p timeout 300 python bench.py --no-cpu-baseline --no-extra-legs --designs $n --steps 6 --warmup 2 --resident 2>/dev/null | tail -1 | python /tmp/_row.py persist=$p $n
  done
done | tee $OUT/nscale.txt
for p in 0 1 0 1; do
  RAFTX_PERSIST=$p timeout 600 python bench.py --no-cpu-baseline --no-extra-legs 2>$OUT/bench_p$p.err | tail -1 > $OUT/bench_p$p.json
  python - $OUT/bench_p$p.json $p <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print('persist=%s' % sys.argv[2], 'value', d['value'], 'ms_per_step', d['ms_per_step'], 'roofline', d['roofline'].get('frac'), 'kernel_ms', d.get('kernel_ms_per_step'))
PY
done | tee $OUT/bench_ab.txt
