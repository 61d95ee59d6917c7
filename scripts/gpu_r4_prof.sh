#!/bin/bash
# Round-4 profiling visit: QTF parity + kernel stats + counters, xi-out copy/kernel timeline.
set -u
TAG=${1:-r04_prof}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
( timeout 600 python -m pytest tests/test_hip_qtf.py tests/test_full_size.py -m gpu -x -q 2>&1 | tail -5 ) > $OUT/pytest.log
cat $OUT/pytest.log
bash scripts/gpu_qtf_prof.sh $TAG/qtf 2>&1 | tail -60
bash scripts/gpu_xi_trace.sh $TAG/xi 2>&1 | tail -40
