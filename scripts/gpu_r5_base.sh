#!/bin/bash
# round-5 baseline visit: parity suite + the default bench line (with the reference-numpy cpu_baseline leg). Usage: bash scripts/gpu_r5_base.sh <tag>
TAG=${1:-r05_base}
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; cd $R
nproc > $OUT/host.txt; free -g >> $OUT/host.txt; lscpu | head -20 >> $OUT/host.txt
( timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -25 ) > $OUT/pytest_gpu.log
( time timeout 900 python bench.py 2>$OUT/bench.err | tail -1 ) > $OUT/bench.json 2> $OUT/bench.time
cat $OUT/host.txt $OUT/pytest_gpu.log; tail -5 $OUT/bench.err; cat $OUT/bench.time; python - <<PY
import json
d=json.loads(open("$OUT/bench.json").read())
print({k:d[k] for k in ("value","ms_per_step")}, d["roofline"]["frac"])
cb=d.get("cpu_baseline",{}); print({k:v for k,v in cb.items() if k not in ("port_simd","details")})
print(d.get("reference_numpy_this_host"))
PY
