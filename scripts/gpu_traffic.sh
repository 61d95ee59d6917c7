#!/bin/bash
# HBM-side traffic counters (separate passes) for a list of shapes. Usage: bash scripts/gpu_traffic.sh <tag> "4,64 1,256"
TAG=${1:-t}; SHAPES=${2:-"4,64 1,256 2,128"}
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp; cd /tmp
for sh in $SHAPES; do
  for c in FETCH_SIZE WRITE_SIZE; do
    d=$OUT/${sh/,/x}_$c
    RAFTX_SHAPE=$sh timeout 300 rocprofv3 --pmc $c --output-format csv -d $d -o b -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $d.log 2>&1
    python3 - $d/b_counter_collection.csv $sh $c <<'PY'
import csv,sys
v=[float(r["Counter_Value"]) for r in csv.DictReader(open(sys.argv[1])) if "k_solve_dynamics" in r["Kernel_Name"]]
print(sys.argv[2], sys.argv[3], "%.1f MB/launch (raw counter KB*1e-3)" % (sum(v)/len(v)/1e3))
PY
  done
done
