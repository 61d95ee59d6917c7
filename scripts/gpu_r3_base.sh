#!/bin/bash
# Round-3 kernel visit: [parity subset] + kernel-time A/B of library variants (exp_timing) + phase timing of the timing build.
# Usage: bash scripts/gpu_r3_base.sh <tag> [variant.so ...]
TAG=${1:-r3}; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; cd $R
( timeout 900 python -m pytest tests/test_hip_parity.py tests/test_full_size.py tests/test_hip_shapes.py -m gpu -x -q 2>&1 | tail -8 ) > $OUT/pytest.log
cat $OUT/pytest.log
( EXP_REPS=2 timeout 600 python scripts/exp_timing.py head "$@" 2>&1 | tail -12 ) | tee $OUT/exp_timing.txt
if [ -f raft_amd/csrc/libraftx_hip_timing.so ]; then ( timeout 300 python scripts/phase_timing.py 2>&1 | tail -12 ) | tee $OUT/phase.txt; fi
