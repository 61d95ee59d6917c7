#!/bin/bash
# Kernel + copy timeline of the streamed crossing with the responses downloaded (--xi-out).
set -u
TAG=${1:-xitrace}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --memory-copy-trace --stats --output-format csv -d $OUT/trace -o bench -- python $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --xi-out > $OUT/trace.log 2>&1
cd $R
find $OUT -name '*.csv' -size +8M -delete
ls $OUT/trace
