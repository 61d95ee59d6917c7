#!/bin/bash
# copy + kernel timeline of the xi-out LEG of a default bench.py run (after the headline), download stream high / low
set -u
TAG=${1:-r04_xi5}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
for P in high low; do
cd /tmp; export TMPDIR=/tmp
RAFTX_BENCH_XI_STEPS=12 RAFTX_D2H_PRIORITY=$P timeout 600 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $OUT/trace_$P -o bench -- python $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --legs xi > $OUT/trace_$P.log 2>&1
cd $R
python - <<PY
import csv, json
d = "$OUT/trace_$P/"
ev = []
for r in csv.DictReader(open(d + "bench_memory_copy_trace.csv")):
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if e - s > 100000: ev.append((s, e, r["Direction"][12:] + " s" + r["Stream_Id"]))
for r in csv.DictReader(open(d + "bench_kernel_trace.csv")):
    if "k_solve_dynamics" in r["Kernel_Name"] or "k_geom_member" in r["Kernel_Name"] or "k_geom_design(" in r["Kernel_Name"]:
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:22] + " q" + r.get("Queue_Id", "?") + " s" + r.get("Stream_Id", "?")))
ev.sort()
d2h = [i for i, e in enumerate(ev) if e[2].startswith("DEVICE_TO_HOST") and e[1] - e[0] > 2500000]
i0 = d2h[len(d2h) // 2 + 2]
t0 = ev[i0][0]
out = ["%9.1f .. %9.1f (%7.1f) %s" % ((s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, n) for s, e, n in ev[i0 - 4: i0 + 40]]
open("$OUT/timeline_$P.txt", "w").write("\n".join(out))
print("==== $P"); print("\n".join(out))
print(open("$OUT/trace_$P.log").read()[-600:][:300])
PY
find $OUT -name '*.csv' -size +8M -delete
done
