#!/bin/bash
# Round 6, third batch: (1) GPU suite with the two-lanes-per-row coupled solver in, (2) that solver against the one-lane form
# (bit-identical? how much faster?), (3) the whole bench -- every leg -- for the default build and for the build with
# machine-LICM off, (4) grid size of the persistent launch at the 1 250-design shard.
TAG=${1:-r06_batch3}
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; cd $R
V=$R/raft_amd/csrc
( timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 ) > $OUT/pytest_gpu.log
cat $OUT/pytest_gpu.log
( RAFTX_HIP_LIB=$V/libraftx_hip_v_nolicm.so timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 ) > $OUT/pytest_gpu_nolicm.log
cat $OUT/pytest_gpu_nolicm.log
for rep in 1 2; do
  for r2 in 0 1; do
    RAFTX_SYSROWS2=$r2 timeout 600 python scripts/bench_farm.py --sweep 1000 2>$OUT/farm_$r2.err | tail -1 > $OUT/farm_$r2.json
    python - $OUT/farm_$r2.json $r2 <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
s = d["farm_sweep"]
print("rows2=%s" % sys.argv[2], "coupled_ms %.3f" % s["coupled_solves_kernel_ms"], "TF %.2f" % s["coupled_solve_tflops"], "units_ms %.3f" % s["unit_fixed_points_kernel_ms"],
      "c4 coupled_ms %.4f" % d["coupled_24x24_solves_kernel_ms"], "err %.2e" % d["max_group_rel_err_vs_live_reference_all_50_sea_states"], flush=True)
PY
  done
done | tee $OUT/farm_ab.txt
python - <<'PY' | tee $OUT/farm_bits.txt
# the two solvers on the same systems: the same bits?
import os, subprocess, sys, json
code = r'''
import numpy as np, sys, os
sys.path.insert(0, os.getcwd())
from raft_amd import backend, dropin
from raft_amd.snapshot import load_model_fixture, case_from_fixture
fx, model = load_model_fixture("c4_farm.npz")
cases = [case_from_fixture(c) for c in fx["cases"]]
sweep = dropin.sweep_from_units(model, cases)
out = sweep.run_farm(backend.default_context(0), 4, Cc=fx["coupling_C"][None])
np.save(sys.argv[1], out["Xi"])
'''
for r2 in ("0", "1"):
    subprocess.check_call([sys.executable, "-c", code, "/tmp/xi_%s.npy" % r2], env=dict(os.environ, RAFTX_SYSROWS2=r2))
import numpy as np
a, b = np.load("/tmp/xi_0.npy"), np.load("/tmp/xi_1.npy")
print("two-lanes-per-row vs one-lane: bit-identical =", bool(np.array_equal(a.view(np.uint64), b.view(np.uint64))), "max abs diff", float(np.abs(a - b).max()))
PY
RAFTX_BENCH_XI_STEPS=30 timeout 900 python bench.py --no-cpu-baseline 2>$OUT/bench_default.err | tail -1 > $OUT/bench_default.json
RAFTX_BENCH_XI_STEPS=30 RAFTX_HIP_LIB=$V/libraftx_hip_v_nolicm.so timeout 900 python bench.py --no-cpu-baseline 2>$OUT/bench_nolicm.err | tail -1 > $OUT/bench_nolicm.json
RAFTX_BENCH_XI_STEPS=30 timeout 900 python bench.py --no-cpu-baseline 2>$OUT/bench_default2.err | tail -1 > $OUT/bench_default2.json
RAFTX_BENCH_XI_STEPS=30 RAFTX_HIP_LIB=$V/libraftx_hip_v_nolicm.so timeout 900 python bench.py --no-cpu-baseline 2>$OUT/bench_nolicm2.err | tail -1 > $OUT/bench_nolicm2.json
python - $OUT <<'PY' | tee $OUT/bench_ab.txt
import json, sys, os
def flat(d, pre=""):
    out = {}
    for k, v in d.items():
        if isinstance(v, dict):
            out.update(flat(v, pre + k + "."))
        elif isinstance(v, (int, float)) and not isinstance(v, bool):
            out[pre + k] = v
    return out
names = ["bench_default", "bench_nolicm", "bench_default2", "bench_nolicm2"]
D = [flat(json.load(open(os.path.join(sys.argv[1], n + ".json")))) for n in names]
keys = [k for k in D[0] if any(t in k for t in ("ms", "frac", "value", "dcf", "tflops")) and all(k in d for d in D)]
for k in keys:
    v = [d[k] for d in D]
    if v[0]:
        print("%-90s %s   nolicm/default %.3f" % (k[:90], " ".join("%.4g" % x for x in v), (v[1] + v[3]) / (v[0] + v[2])))
PY
cat > /tmp/_row.py <<'PY'
import json, sys
d = json.loads(sys.stdin.read())
r = d['roofline']
print(' '.join(sys.argv[1:]), 'ms_per_step %.4f' % d['ms_per_step'], 'kernel %.4f' % r['kernel_ms_per_step'], flush=True)
PY
for g in 0 512 640 768 896; do
  RAFTX_KP_GRID=$g timeout 300 python bench.py --no-cpu-baseline --no-extra-legs --designs 1250 --steps 60 --warmup 5 2>/dev/null | tail -1 | python /tmp/_row.py n=1250 grid=$g
done | tee $OUT/kp_grid_1250.txt
