#!/bin/bash
# Instruction counters of the fused kernel on the bench workload (one whole-batch launch per step) + phase timing.
# Usage: bash scripts/gpu_r3_pmc.sh <tag>
TAG=${1:-r3pmc}
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; cd $R
export TMPDIR=/tmp
( EXP_REPS=2 timeout 600 python scripts/exp_timing.py head 2>&1 | tail -3 ) | tee $OUT/exp_timing.txt
if [ -f raft_amd/csrc/libraftx_hip_timing.so ]; then ( timeout 300 python scripts/phase_timing.py 2>&1 | tail -12 ) | tee $OUT/phase.txt; fi
cd /tmp
BENCH="python $R/scripts/exp_timing.py head"
EXP_REPS=1 timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY --output-format csv -d $OUT/pmc -o x -- $BENCH > $OUT/pmc.log 2>&1
cd $R
python - <<PY | tee $OUT/pmc_summary.txt
import csv, glob, collections
for f in glob.glob("$OUT/pmc/**/*counter_collection.csv", recursive=True):
    acc = collections.defaultdict(list)
    for row in csv.DictReader(open(f)):
        if "k_solve_dynamics" in row["Kernel_Name"]:
            acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
    print({k: "%.4g" % (sum(v) / max(1, len(v))) for k, v in acc.items()}, "launches", {k: len(v) for k, v in acc.items()})
PY
find $OUT -name '*.csv' -size +8M -delete
