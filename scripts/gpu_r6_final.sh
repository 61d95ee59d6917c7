#!/bin/bash
# Round-6 visit: gpu_round.sh (parity suite, smoke, default bench, kernel trace + PMC passes of the C3 sweep) + kernel statistics
# of the farm sweep (C4), the QTF batches (C5) and the flexible sweep.
set -u
TAG=${1:-r06_final}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
bash scripts/gpu_round.sh $TAG > $OUT.round.log 2>&1
tail -12 $OUT.round.log
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/farm_trace -o farm -- python $R/scripts/bench_farm.py --sweep 1000 > $OUT/farm_trace.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/qtf_trace -o qtf -- python $R/scripts/bench_qtf.py > $OUT/qtf_trace.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/flex_trace -o flex -- python $R/scripts/bench_flex.py 16 > $OUT/flex_trace.log 2>&1
cd $R
find $OUT -name '*_kernel_trace.csv' -size +4M -delete
cut -c1-170 $OUT/farm_trace/farm_kernel_stats.csv | head -6
cut -c1-170 $OUT/qtf_trace/qtf_kernel_stats.csv | head -8
cut -c1-170 $OUT/flex_trace/flex_kernel_stats.csv | head -8
