#!/bin/bash
# Tuning visit for the sweep crossing: geometry / crossing tests, then bench.py under several block splits.
# Usage: bash scripts/gpu_cross.sh <tag>
set -u
TAG=${1:-cross}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
( timeout 600 python -m pytest tests/test_geometry.py tests/test_full_size.py -m gpu -x -q 2>&1 | tail -15 ) > $OUT/pytest.log
for split in "default" "1" "0.2,0.8" "0.1,0.9" "0.1,0.3,0.6" "0.08,0.22,0.7" "0.1,0.2,0.3,0.4" "0.25,0.25,0.25,0.25"; do
  if [ "$split" = "default" ]; then unset RAFTX_SWEEP_SPLIT; else export RAFTX_SWEEP_SPLIT=$split; fi
  echo "== split $split" >> $OUT/bench.log
  ( timeout 300 python bench.py --no-cpu-baseline --steps 20 --warmup 3 2>&1 | tail -1 | python -c '
import sys, json
l = sys.stdin.read().strip()
try:
    d = json.loads(l)
    print("value %.1f M  ms/step %.3f  breakdown %s  frac64 %.3f" % (d["value"] / 1e6, d["ms_per_step"], d["step_breakdown_ms"], d["roofline_fp64_valu"]["frac"]))
except Exception as e:
    print("FAILED", l[-2000:])
' ) >> $OUT/bench.log
done
unset RAFTX_SWEEP_SPLIT
( timeout 300 python bench.py --no-cpu-baseline --steps 20 --warmup 3 --xi-out 2>&1 | tail -1 | cut -c1-400 ) > $OUT/bench_xi.log
cat $OUT/pytest.log $OUT/bench.log $OUT/bench_xi.log
