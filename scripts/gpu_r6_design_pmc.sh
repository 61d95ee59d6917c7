#!/bin/bash
# Round 6: where k_geom_design's wave time goes (sweep form: no ABI copy), counters of the geometry kernels inside the streamed bench
TAG=${1:-r06_design_pmc}
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
B="python $R/bench.py --steps 6 --warmup 2 --profile --no-cpu-baseline --no-extra-legs"
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA --output-format csv -d $OUT/p1 -o b -- $B > $OUT/p1.log 2>&1
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS --output-format csv -d $OUT/p2 -o b -- $B > $OUT/p2.log 2>&1
timeout 300 rocprofv3 --pmc SQ_INST_CYCLES_SMEM SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT SQ_ACTIVE_INST_MISC SQ_INSTS_BRANCH SQ_IFETCH SQ_INSTS_EXP_GDS --output-format csv -d $OUT/p3 -o b -- $B > $OUT/p3.log 2>&1
cd $R
python - $OUT <<'PY'
import csv, sys, os, glob
from collections import defaultdict
for sub in ("p1", "p2", "p3"):
    for p in glob.glob(os.path.join(sys.argv[1], sub, "*counter_collection.csv")):
        acc = defaultdict(lambda: defaultdict(float)); cnt = defaultdict(lambda: defaultdict(int))
        for r in csv.DictReader(open(p)):
            k = r["Kernel_Name"][:24]
            if not (k.startswith("k_geom_design") or k.startswith("k_geom_member") or k.startswith("k_motion")): continue
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[k][r["Counter_Name"]] += 1
        for k, v in acc.items():
            print(sub, "%-24s" % k, " ".join("%s=%.4g" % (c, x / max(cnt[k][c], 1)) for c, x in sorted(v.items())))
PY
