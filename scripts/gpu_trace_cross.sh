#!/bin/bash
# Kernel + copy timeline of the sweep crossing.  Usage: bash scripts/gpu_trace_cross.sh <tag> [split]
set -u
TAG=${1:-trace}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
[ -n "${2:-}" ] && export RAFTX_SWEEP_SPLIT=$2
cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --memory-copy-trace --stats --output-format csv -d $OUT/trace -o bench -- python $R/bench.py --steps 6 --warmup 3 --no-cpu-baseline > $OUT/trace.log 2>&1
cd $R
find $OUT -name '*.csv' -size +8M -delete
ls $OUT/trace; tail -2 $OUT/trace.log | cut -c1-600
