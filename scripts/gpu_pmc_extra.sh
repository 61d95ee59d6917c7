#!/bin/bash
# Extra SQ counters of the fused kernel (instruction fetch, scalar / vector memory levels).  Usage: bash scripts/gpu_pmc_extra.sh <tag>
TAG=${1:-pmcx}
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
BENCH="python $R/scripts/exp_timing.py head"
for set in "SQ_IFETCH SQ_IFETCH_LEVEL SQ_INSTS_SMEM SQ_INST_LEVEL_SMEM SQ_INSTS_BRANCH SQ_WAVE_CYCLES" "SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM SQ_INST_CYCLES_SALU SQ_INST_CYCLES_SMEM SQ_INST_LEVEL_LDS SQ_INSTS_LDS SQ_CYCLES"; do
  n=$(echo $set | cut -d' ' -f1)
  EXP_REPS=1 timeout 300 rocprofv3 --pmc $set --output-format csv -d $OUT/$n -o x -- $BENCH > $OUT/$n.log 2>&1
done
cd $R
python - <<PY | tee $OUT/summary.txt
import csv, glob, collections
acc = collections.defaultdict(list)
for f in glob.glob("$OUT/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "k_solve_dynamics" in row["Kernel_Name"]:
            acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
print({k: "%.4g" % (sum(v) / max(1, len(v))) for k, v in acc.items()})
PY
find $OUT -name '*.csv' -size +8M -delete
