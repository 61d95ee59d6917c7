#!/bin/bash
# Instruction-fetch counters of the main kernel. Usage: bash scripts/gpu_icache.sh <tag>
TAG=${1:-ic}
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp; cd /tmp
rocprofv3 --list-avail > $OUT/avail.txt 2>&1
grep -i -o "SQC_[A-Z_0-9]*\|SQ_IFETCH[A-Z_0-9]*\|SQ_WAIT_IFETCH[A-Z_0-9]*\|SQ_INST_LEVEL[A-Z_0-9]*\|SQ_INSTS_BRANCH[A-Z_0-9]*\|SQ_ACTIVE_INST[A-Z_0-9]*" $OUT/avail.txt | sort -u > $OUT/names.txt
cat $OUT/names.txt | tr '\n' ' '
BENCH="python $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline"
i=0
for grp in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE" "SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAVE_CYCLES SQ_WAIT_INST_ANY" "SQ_INSTS_BRANCH SQ_INSTS_CBRANCH_NOT_TAKEN SQ_INSTS_CBRANCH_TAKEN SQ_BUSY_CYCLES"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $grp --output-format csv -d $OUT/pmc$i -o bench -- $BENCH > $OUT/pmc$i.log 2>&1
  tail -3 $OUT/pmc$i.log
done
cd $R
python - <<PY
import csv, glob, collections
for d in sorted(glob.glob("$OUT/pmc*/")):
    for f in glob.glob(d + "**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(list)
        for row in csv.DictReader(open(f)):
            if "k_solve_dynamics" in row["Kernel_Name"]:
                acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
        for k, v in acc.items():
            print(k, "per launch: %.4g" % (sum(v) / max(1, len(v))), "n", len(v))
PY
find $OUT -name '*.csv' -size +8M -delete
