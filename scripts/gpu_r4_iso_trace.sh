#!/bin/bash
# copy + kernel timeline of an isolated crossing with the responses out (the last of ISO_CALLS calls of scripts/iso_xi.py)
set -u
TAG=${1:-r04_iso}
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
RAFTX_SWEEP_DEBUG=1 ISO_CALLS=${ISO_CALLS:-6} timeout 300 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $OUT/trace -o iso -- python $R/scripts/iso_xi.py > $OUT/trace.log 2>&1
grep "isolated\|raftx_sweep slot\|median" $OUT/trace.log | tail -6
cd $R
python scripts/trace_timeline.py $OUT/trace/iso | tee $OUT/timeline.txt
find $OUT -name '*.csv' -size +8M -delete
