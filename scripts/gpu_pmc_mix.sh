#!/bin/bash
# VALU instruction mix of the sweep kernel. Usage: bash scripts/gpu_pmc_mix.sh <tag>
TAG=${1:-mix}; R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp; cd /tmp
B="python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --chunks 1"
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_SALU SQ_INSTS_SMEM --output-format csv -d $OUT/mix -o b -- $B > $OUT/mix.log 2>&1
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INST_CYCLES_SMEM SQ_INSTS_LDS --output-format csv -d $OUT/cyc -o b -- $B > $OUT/cyc.log 2>&1
python3 - $OUT <<'PY'
import csv,sys,collections,glob
for d in ("mix","cyc"):
    acc=collections.defaultdict(list)
    for f in glob.glob(sys.argv[1]+"/"+d+"/b_counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            if "k_solve_dynamics" in r["Kernel_Name"]: acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k,v in acc.items(): print(d,k,"%.4g"%(sum(v)/len(v)))
PY
tail -3 $OUT/mix.log
