#!/bin/bash
# configs[3] (farm) and configs[4] (QTF) at their BASELINE shapes: parity tests, bench JSON lines and rocprofv3 kernel stats.
# Usage: bash scripts/gpu_c4c5.sh <tag>
set -u
TAG=${1:-c4c5}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
( timeout 600 python -m pytest tests/test_hip_parity.py tests/test_hip_qtf.py tests/test_full_size.py -m gpu -x -q 2>&1 | tail -6 ) > $OUT/pytest.log
( timeout 300 python scripts/bench_farm.py 2>&1 | tail -1 ) > $OUT/farm.json
( timeout 300 python scripts/bench_qtf.py 2>&1 | tail -3 ) > $OUT/qtf.json
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/farm_trace -o farm -- python $R/scripts/bench_farm.py > $OUT/farm_trace.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/qtf_trace -o qtf -- python $R/scripts/bench_qtf.py > $OUT/qtf_trace.log 2>&1
cd $R
find $OUT -name '*_kernel_trace.csv' -size +4M -delete
cat $OUT/pytest.log $OUT/farm.json $OUT/qtf.json
cut -c1-150 $OUT/farm_trace/farm_kernel_stats.csv | head -8
cut -c1-150 $OUT/qtf_trace/qtf_kernel_stats.csv | head -10
