#!/bin/bash
# LDS counters of k_geom_design (is the LDS pipe what bounds it?)
cd /tmp; export TMPDIR=/tmp
rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_BUSY_CYCLES SQ_WAVE_CYCLES --output-format csv -d /tmp/dl -o g -- python $GRAFT_REPO_ROOT/scripts/bench_geom.py --reps 3 > /dev/null 2>&1
python - <<'PY'
import csv, glob
from collections import defaultdict
for p in glob.glob("/tmp/dl/*counter_collection.csv"):
    acc = defaultdict(lambda: defaultdict(float)); cnt = defaultdict(lambda: defaultdict(int))
    for r in csv.DictReader(open(p)):
        k = r["Kernel_Name"][:22]
        if not k.startswith("k_geom_design"): continue
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[k][r["Counter_Name"]] += 1
    for k, v in acc.items():
        print(k, " ".join("%s=%.4g" % (c, x / cnt[k][c]) for c, x in sorted(v.items())))
PY
