#!/bin/bash
# xi-out A/B: download stream in the lowest / highest priority class; staged three-deep loop; copy + kernel timeline.
set -u
TAG=${1:-r04_xi}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
for P in low high; do
  ( RAFTX_D2H_PRIORITY=$P timeout 300 python bench.py --no-cpu-baseline 2>$OUT/bench_$P.err | tail -1 ) > $OUT/bench_$P.json
  python - <<PY
import json
d = json.load(open("$OUT/bench_$P.json"))
print("$P: value %.1f M ms/step %.3f kernel %.3f | xi_out %s" % (d["value"]/1e6, d["ms_per_step"], d["roofline"]["kernel_ms_per_step"], {k: round(v, 3) for k, v in d["xi_out"].items() if k.endswith("ms_per_step")}))
PY
done

bash scripts/gpu_xi_trace.sh $TAG/trace 2>&1 | tail -5
python - <<PY
import csv
d = "$OUT/trace/trace/"
ev = []
for r in csv.DictReader(open(d + "bench_memory_copy_trace.csv")):
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if e - s > 100000: ev.append((s, e, r["Direction"][12:]))
for r in csv.DictReader(open(d + "bench_kernel_trace.csv")):
    if "k_solve_dynamics" in r["Kernel_Name"] or "k_geom_member" in r["Kernel_Name"] or "k_geom_design(" in r["Kernel_Name"]:
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:22]))
ev.sort()
t0 = ev[len(ev) // 2][0]
out = ["%9.1f .. %9.1f (%7.1f) %s" % ((s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, n) for s, e, n in ev[len(ev) // 2: len(ev) // 2 + 36]]
open("$OUT/xi_timeline.txt", "w").write("\n".join(out))
print("\n".join(out))
PY
