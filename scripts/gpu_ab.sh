#!/bin/bash
# A/B of library variants on one box: bench (3 repeats each, interleaved) + VALU instruction counts.
# Usage: bash scripts/gpu_ab.sh <tag> <variant.so> [<variant2.so> ...]   (the product library is always leg "head")
TAG=${1:-ab}; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; cd $R
export TMPDIR=/tmp
one() { python bench.py --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["roofline"]["kernel_ms"])'; }
for rep in 1 2 3; do
  echo "head $(one)" | tee -a $OUT/ab.txt
  for v in "$@"; do
    echo "$(basename $v) $(RAFTX_HIP_LIB=$R/$v one)" | tee -a $OUT/ab.txt
  done
done
cd /tmp
BENCH="python $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline"
PMC="SQ_INSTS_VALU SQ_INSTS_SALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES"
timeout 300 rocprofv3 --pmc $PMC --output-format csv -d $OUT/pmc_head -o bench -- $BENCH > $OUT/pmc_head.log 2>&1
for v in "$@"; do
  RAFTX_HIP_LIB=$R/$v timeout 300 rocprofv3 --pmc $PMC --output-format csv -d $OUT/pmc_$(basename $v .so) -o bench -- $BENCH > $OUT/pmc_$(basename $v .so).log 2>&1
done
cd $R
python - <<PY | tee -a $OUT/ab.txt
import csv, glob, collections
for d in sorted(glob.glob("$OUT/pmc_*/")):
    for f in glob.glob(d + "**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(list)
        for row in csv.DictReader(open(f)):
            if "k_solve_dynamics" in row["Kernel_Name"]:
                acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
        print(d.split("/")[-2], {k: "%.4g" % (sum(v) / max(1, len(v))) for k, v in acc.items()})
PY
find $OUT -name '*.csv' -size +8M -delete
