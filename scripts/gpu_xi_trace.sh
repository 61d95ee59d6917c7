#!/bin/bash
# Kernel + memory-copy timeline of streamed steps with the responses downloaded (--xi-out).  Usage: bash scripts/gpu_xi_trace.sh <tag>
TAG=${1:-xi}
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $OUT/trace -o bench -- python $R/bench.py --steps 8 --warmup 3 --profile --xi-out --no-extra-legs > $OUT/trace.log 2>&1
cd $R
python - <<PY
import csv, glob
kf = glob.glob("$OUT/trace/**/*kernel_trace.csv", recursive=True)[0]
mf = glob.glob("$OUT/trace/**/*memory_copy_trace.csv", recursive=True)
ev = []
for r in csv.DictReader(open(kf)):
    n = r["Kernel_Name"]
    if "k_solve_dynamics" in n or "copyBuffer" in n or "k_geom_design(" in n:
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), n[:32]))
if mf:
    for r in csv.DictReader(open(mf[0])):
        b = int(r.get("Bytes", r.get("Size", 0)) or 0)
        if b > 1 << 20:
            ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY %s %.0f MB" % (r.get("Direction", r.get("Name", "?")), b / 1e6)))
ev.sort()
t0 = ev[len(ev) // 2][0]
for s, e, n in ev[len(ev) // 2: len(ev) // 2 + 28]:
    print("%9.1f us .. %9.1f us  (%7.1f)  %s" % ((s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, n))
PY
find $OUT -name '*.csv' -size +8M -delete
