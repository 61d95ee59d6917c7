#!/bin/bash
# kernel-trace statistics of a short streamed bench run. Usage: bash scripts/gpu_ktrace.sh <tag> [bench args]
TAG=${1:-kt}; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o bench -- python $R/bench.py --steps 30 --warmup 5 --no-cpu-baseline "$@" > $OUT/trace.log 2>&1
cut -c1-150 $OUT/trace/bench_kernel_stats.csv | head -16
tail -1 $OUT/trace.log | cut -c1-300
find $OUT -name '*.csv' -size +8M -delete
