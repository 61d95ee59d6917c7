#!/bin/bash
# Kernel timeline of streamed steps: what runs between consecutive k_solve_dynamics launches.  Usage: bash scripts/gpu_timeline.sh <tag>
TAG=${1:-tl}
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -o bench -- python $R/bench.py --steps 12 --warmup 3 --profile > $OUT/trace.log 2>&1
cd $R
python - <<PY
import csv, glob
f = glob.glob("$OUT/trace/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
ks = [r for r in rows if "k_solve_dynamics" in r["Kernel_Name"]]
sel = ks[-6:-1]
for a, b in zip(sel, sel[1:]):
    e, s = int(a["End_Timestamp"]), int(b["Start_Timestamp"])
    print("--- fused end -> next fused start: %.1f us (fused %.1f us)" % ((s - e) / 1e3, (int(a["End_Timestamp"]) - int(a["Start_Timestamp"])) / 1e3))
    for r in rows:
        st, en = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        if en > e - 600000 and st < s + 1000 and "k_solve_dynamics" not in r["Kernel_Name"]:
            print("   %-28s start %+8.1f us  end %+8.1f us  (queue %s)" % (r["Kernel_Name"][:28], (st - e) / 1e3, (en - e) / 1e3, r.get("Queue_Id", "?")))
PY
find $OUT -name '*.csv' -size +8M -delete
