#!/bin/bash
set -u
TAG=${1:-r04_iso}
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
RAFTX_SWEEP_DEBUG=1 timeout 300 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $OUT/trace -o iso -- python $R/scripts/iso_xi.py > $OUT/trace.log 2>&1
grep "isolated\|raftx_sweep slot" $OUT/trace.log | tail -8
cd $R
python - <<PY
import csv
d = "$OUT/trace/"
ev = []
for r in csv.DictReader(open(d + "iso_memory_copy_trace.csv")):
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if e - s > 30000: ev.append((s, e, r["Direction"][12:]))
for r in csv.DictReader(open(d + "iso_kernel_trace.csv")):
    n = r["Kernel_Name"]
    if n.startswith("k_") or "k_solve" in n: ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), n[:26]))
ev.sort()
last = [i for i, e in enumerate(ev) if e[2].startswith("HOST_TO_DEVICE")]
# the last call: from the first H2D after the previous call's last D2H
d2h = [i for i, e in enumerate(ev) if e[2].startswith("DEVICE_TO_HOST")]
end_prev = d2h[-6] if len(d2h) > 6 else 0
i0 = min(i for i in last if i > end_prev)
t0 = ev[i0][0]
out = ["%9.1f .. %9.1f (%7.1f) %s" % ((s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, n) for s, e, n in ev[i0:]]
open("$OUT/timeline.txt", "w").write("\n".join(out)); print("\n".join(out))
PY
find $OUT -name '*.csv' -size +8M -delete
