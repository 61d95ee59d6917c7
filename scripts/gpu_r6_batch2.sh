#!/bin/bash
# Round 6, second batch of same-box A/Bs:
#  1. the new GPU tests (stale-Z order, moorMod == 2 + internal QTF) and the whole GPU suite with the handover code built in
#  2. machine-LICM off (variant library) against the default build: fused kernel resident, 10 000 pairs
#  3. XiLast in LDS at three pairs per CU (variant library) against the scratch form at four: 10 000 / 40 000 pairs + traffic
#  4. handover of consecutive persistent grids (RAFTX_HANDOVER=1) with 0 / 32 / 64 / 128 reserved places, depth 3 / 4
TAG=${1:-r06_batch2}
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; cd $R
( timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 ) > $OUT/pytest_gpu.log
cat $OUT/pytest_gpu.log
cat > /tmp/_res.py <<'PY'
import json, sys
d = json.loads(sys.stdin.read())
k = d['kernel_resident']
t = d['roofline'].get('traffic')
print(' '.join(sys.argv[1:]), 'kernel_ms %.4f' % k['kernel_ms'], 'mean_it %.3f' % d['mean_iterations'], 'traffic_GB', None if not t else round(t / 1e9, 3), flush=True)
PY
V=$R/raft_amd/csrc
for rep in 1 2 3; do
  timeout 300 python bench.py --no-cpu-baseline --no-extra-legs --designs 10000 --steps 6 --warmup 2 --resident 2>/dev/null | tail -1 | python /tmp/_res.py lib=default n=10000
  RAFTX_HIP_LIB=$V/libraftx_hip_v_nolicm.so timeout 300 python bench.py --no-cpu-baseline --no-extra-legs --designs 10000 --steps 6 --warmup 2 --resident 2>/dev/null | tail -1 | python /tmp/_res.py lib=nolicm n=10000
done | tee $OUT/nolicm.txt
for n in 10000 40000; do
  for rep in 1 2; do
  timeout 300 python bench.py --no-cpu-baseline --legs traffic --designs $n --steps 6 --warmup 2 --resident 2>/dev/null | tail -1 | python /tmp/_res.py lib=default xl=scratch pairs_per_cu=4 n=$n
  RAFTX_HIP_LIB=$V/libraftx_hip_v_xllds.so RAFTX_WG_PER_CU=3 timeout 300 python bench.py --no-cpu-baseline --legs traffic --designs $n --steps 6 --warmup 2 --resident 2>$OUT/xllds_$n.err | tail -1 | python /tmp/_res.py lib=xllds xl=lds pairs_per_cu=3 n=$n
  done
done | tee $OUT/xilast.txt
cat > /tmp/_row.py <<'PY'
import json, sys
d = json.loads(sys.stdin.read())
r = d['roofline']
print(' '.join(sys.argv[1:]), 'ms_per_step %.4f' % d['ms_per_step'], 'value %.1fM' % (d['value'] / 1e6), 'kernel_union %.4f' % r['kernel_ms_per_step'],
      'per_launch %.4f' % r['kernel_ms_per_launch'], 'frac %.4f' % r['frac'], 'step_frac %.4f' % r['step_frac'], flush=True)
PY
for n in 10000 1250; do
  timeout 300 python bench.py --no-cpu-baseline --no-extra-legs --designs $n --depth 3 --steps 40 --warmup 5 2>/dev/null | tail -1 | python /tmp/_row.py n=$n handover=0 depth=3
  for rs in 0 32 64 128; do
    for dp in 3 4; do
      RAFTX_HANDOVER=1 RAFTX_KP_RESERVE=$rs timeout 300 python bench.py --no-cpu-baseline --no-extra-legs --designs $n --depth $dp --steps 40 --warmup 5 2>$OUT/err_${n}_${rs}_${dp}.txt | tail -1 | python /tmp/_row.py n=$n handover=1 reserve=$rs depth=$dp
    done
  done
done | tee $OUT/handover.txt
cd /tmp; export TMPDIR=/tmp
RAFTX_HANDOVER=1 RAFTX_KP_RESERVE=64 timeout 600 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -o bench -- python $R/bench.py --no-cpu-baseline --no-extra-legs --depth 3 --steps 12 --warmup 3 --profile > $OUT/trace.log 2>&1
cd $R
find $OUT -name '*.csv' -size +8M -delete
