#!/bin/bash
# quick GPU visit: parity suite + bench (no profiling). Usage: bash scripts/gpu_quick.sh <tag> [bench args]
TAG=${1:-q}; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; cd $R
( timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -25 ) > $OUT/pytest_gpu.log
( timeout 600 python bench.py --no-cpu-baseline "$@" 2>$OUT/bench.err | tail -1 ) > $OUT/bench.json
cat $OUT/pytest_gpu.log; tail -5 $OUT/bench.err; cat $OUT/bench.json
