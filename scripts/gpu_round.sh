#!/bin/bash
# One GPU-box visit: parity suite, smoke, bench, rocprofv3 kernel trace + PMC passes.
# Usage (from the repo root on the GPU box): bash scripts/gpu_round.sh <tag>
set -u
TAG=${1:-r01}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
( timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 ) > $OUT/pytest_gpu.log
( timeout 300 python -c 'import __graft_entry__ as g; g.smoke()' 2>&1 | tail -5 ) > $OUT/smoke.log
( timeout 600 python bench.py --resident 2>$OUT/bench.err | tail -1 ) > $OUT/bench.json
# counter passes: one launch of the fused kernel per step (--chunks 1), so that per-launch counters are per-step counters
BENCH="python $R/bench.py --steps 5 --warmup 1 --profile"
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o bench -- python $R/bench.py --steps 40 --warmup 10 --profile > $OUT/trace.log 2>&1   # long enough for the clocks to settle: the average then agrees with bench.py
timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o bench -- $BENCH > $OUT/pmc_fetch.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o bench -- $BENCH > $OUT/pmc_write.log 2>&1
timeout 600 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d $OUT/pmc_sq -o bench -- $BENCH > $OUT/pmc_sq.log 2>&1
timeout 600 rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS --output-format csv -d $OUT/pmc_sq2 -o bench -- $BENCH > $OUT/pmc_sq2.log 2>&1
cd $R
find $OUT -name '*.csv' -size +8M -delete
ls -R $OUT | head -60
cat $OUT/pytest_gpu.log $OUT/smoke.log $OUT/bench.json
