#!/bin/bash
# Round 6: instruction counts of the geometry kernels (are they issue-bound or latency-bound?)
TAG=${1:-r06_geom_pmc}
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES --output-format csv -d $OUT/pmc1 -o geom -- python $R/scripts/bench_geom.py --reps 3 > $OUT/pmc1.log 2>&1
timeout 300 rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM --output-format csv -d $OUT/pmc2 -o geom -- python $R/scripts/bench_geom.py --reps 3 > $OUT/pmc2.log 2>&1
cd $R
python - $OUT <<'PY'
import csv, sys, os, glob
from collections import defaultdict
for sub in ("pmc1", "pmc2"):
    for p in glob.glob(os.path.join(sys.argv[1], sub, "*counter_collection.csv")):
        acc = defaultdict(lambda: defaultdict(float)); cnt = defaultdict(int)
        for r in csv.DictReader(open(p)):
            k = r["Kernel_Name"][:28]
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
            if r["Counter_Name"] in ("SQ_WAVES", "SQ_BUSY_CYCLES"): cnt[k] += 1
        for k, v in acc.items():
            n = max(cnt[k], 1)
            print(sub, "%-28s" % k, " ".join("%s=%.3g" % (c, x / n) for c, x in sorted(v.items())))
PY
