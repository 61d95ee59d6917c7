#!/bin/bash
# Kernel iteration visit: parity suite (fast subset or full), then the resident-kernel timing of bench.py.
# Usage: bash scripts/gpu_kernel_check.sh <tag> [full]
set -u
TAG=${1:-kchk}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
if [ "${2:-}" = "full" ]; then
  ( timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -25 ) > $OUT/pytest.log
else
  ( timeout 600 python -m pytest tests/test_hip_parity.py tests/test_full_size.py -m gpu -x -q 2>&1 | tail -25 ) > $OUT/pytest.log
fi
( timeout 300 python bench.py --no-cpu-baseline --steps 20 --warmup 3 --resident 2>&1 | tail -1 | python -c '
import sys, json
l = sys.stdin.read().strip()
try:
    d = json.loads(l)
    print("value %.1f M  ms/step %.3f  solve_sum %.3f  resident %s parity %s" % (d["value"] / 1e6, d["ms_per_step"], d["step_breakdown_ms"]["solve_kernels_sum"], d.get("kernel_resident"), d["parity"]))
except Exception as e:
    print("FAILED", l[-3000:])
' ) > $OUT/bench.log 2>&1
cat $OUT/pytest.log $OUT/bench.log
