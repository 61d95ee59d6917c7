#!/bin/bash
# where the fused kernel's issue slots go: instruction-class activity, scalar / LDS / fetch levels, lane utilisation
set -u
TAG=${1:-r04_stall}
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
BENCH="python $R/bench.py --steps 5 --warmup 1 --profile"
timeout 300 rocprofv3 --pmc SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_SALU SQ_INST_CYCLES_SMEM SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_WAVE_CYCLES --output-format csv -d $OUT/p1 -o bench -- $BENCH > $OUT/p1.log 2>&1
timeout 300 rocprofv3 --pmc SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT SQ_IFETCH SQ_IFETCH_LEVEL SQ_INST_LEVEL_SMEM SQ_INST_LEVEL_LDS --output-format csv -d $OUT/p2 -o bench -- $BENCH > $OUT/p2.log 2>&1
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CU_CYCLES SQ_CYCLES --output-format csv -d $OUT/p3 -o bench -- $BENCH > $OUT/p3.log 2>&1
cd $R
python - <<PY
import csv, glob, collections, json
out = {}
for f in glob.glob("$OUT/p*/**/*counter_collection.csv", recursive=True):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "k_solve_dynamics" in r["Kernel_Name"]: acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in acc.items(): out[k] = sum(v) / len(v)
json.dump(out, open("$OUT/stall_counters.json", "w"), indent=1)
for k in sorted(out): print("%-28s %.4g" % (k, out[k]))
PY
find $OUT -name '*.csv' -size +8M -delete
