#!/bin/bash
# full GPU parity suite + smoke + default bench
set -u
TAG=${1:-full}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
( timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -6 ) | tee $OUT/pytest_gpu.log
( timeout 300 python -c 'import __graft_entry__ as g; g.smoke()' 2>&1 | tail -2 ) | tee $OUT/smoke.log
bash scripts/gpu_bench_default.sh $TAG
