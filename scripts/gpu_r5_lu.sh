#!/bin/bash
# LU kernels after the two-phase pivot search: parity tests that touch them + the farm / flexible legs of the bench. Usage: bash scripts/gpu_r5_lu.sh <tag>
TAG=${1:-r05_lu}
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; cd $R
( timeout 900 python -m pytest tests -m gpu -x -q -k "system or farm or flex or dense or c4 or array_level" 2>&1 | tail -8 ) > $OUT/pytest_lu.log
( timeout 600 python bench.py --no-cpu-baseline --legs configs --steps 5 --warmup 2 2>$OUT/bench.err | tail -1 ) > $OUT/bench.json
cat $OUT/pytest_lu.log; tail -3 $OUT/bench.err; python - <<PY
import json
d=json.loads(open("$OUT/bench.json").read())
print("c4", d["c4_farm"]["as_specified"], d["c4_farm"]["farm_sweep"])
print("flex", d.get("flex_sweep"))
PY
