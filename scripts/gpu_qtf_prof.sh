#!/bin/bash
# rocprofv3 evidence for the QTF kernels (configs[4] shapes): kernel stats and the fp64 instruction mix of k_qtf_pairs.
set -u
TAG=${1:-qtfprof}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o qtf -- python $R/scripts/bench_qtf.py > $OUT/trace.log 2>&1
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_SALU SQ_WAVES --output-format csv -d $OUT/mix -o qtf -- python $R/scripts/bench_qtf.py > $OUT/mix.log 2>&1
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_INSTS_LDS --output-format csv -d $OUT/cyc -o qtf -- python $R/scripts/bench_qtf.py > $OUT/cyc.log 2>&1
cd $R
python3 - $OUT <<'PY'
import csv,sys,collections,glob,json
out={}
for d in ("mix","cyc"):
    acc=collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(sys.argv[1]+"/"+d+"/qtf_counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            k=r["Kernel_Name"].split("(")[0]
            if k.startswith("k_qtf") or k.startswith("k_kay"):
                acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k,v in acc.items():
        out.setdefault(k,{}).update({c:max(x) for c,x in v.items()})      # the largest launch of each kernel (the 16-set one)
json.dump(out,open(sys.argv[1]+"/pmc_qtf.json","w"),indent=1)
print(json.dumps(out,indent=1))
PY
cut -c1-150 $OUT/trace/qtf_kernel_stats.csv | head -12
