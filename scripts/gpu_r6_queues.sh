#!/bin/bash
# Round 6: hardware queues.  The ordinary streams of a process share GPU_MAX_HW_QUEUES (4) hardware queues, in order each: a
# small kernel that cannot get onto the chip (a persistent grid fills it) blocks whatever is queued behind it on the same
# hardware queue -- e.g. the next batch's fused kernel.  More hardware queues = every stream its own.
TAG=${1:-r06_queues}
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; cd $R
cat > /tmp/_row.py <<'PY'
import json, sys
d = json.loads(sys.stdin.read())
r = d['roofline']
print(' '.join(sys.argv[1:]), 'ms_per_step %.4f' % d['ms_per_step'], 'value %.1fM' % (d['value'] / 1e6), 'kernel_union %.4f' % r['kernel_ms_per_step'],
      'per_launch %.4f' % r['kernel_ms_per_launch'], 'frac %.4f' % r['frac'], 'step_frac %.4f' % r['step_frac'], flush=True)
PY
for rep in 1 2; do
for n in 10000 1250; do
  for q in 4 12; do
  for st in 1 2; do
    for dp in 3 4; do
      GPU_MAX_HW_QUEUES=$q RAFTX_SWEEP_STREAMS=$st timeout 300 python bench.py --no-cpu-baseline --no-extra-legs --designs $n --depth $dp --steps 40 --warmup 5 2>$OUT/err_${n}_${q}_${st}_${dp}.txt | tail -1 | python /tmp/_row.py n=$n queues=$q streams=$st depth=$dp
    done
  done
  done
done
done | tee $OUT/ab.txt
for dp in 3 4; do
cd /tmp; export TMPDIR=/tmp
GPU_MAX_HW_QUEUES=12 RAFTX_SWEEP_STREAMS=2 timeout 600 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace$dp -o bench -- python $R/bench.py --no-cpu-baseline --no-extra-legs --depth $dp --steps 12 --warmup 3 --profile > $OUT/trace$dp.log 2>&1
cd $R
python - <<PY | tee $OUT/timeline$dp.txt
import csv, glob
f = glob.glob("$OUT/trace$dp/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
ks = [r for r in rows if "raftx_kp" in r["Kernel_Name"] or "k_solve_dynamics" in r["Kernel_Name"]]
sel = ks[-9:-3]
t0 = int(sel[0]["Start_Timestamp"])
t1 = int(sel[-1]["End_Timestamp"])
for r in rows:
    st, en = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if en >= t0 and st <= t1:
        print("%-30s start %9.1f us  end %9.1f us  dur %8.1f  (queue %s)" % (r["Kernel_Name"][:30], (st - t0) / 1e3, (en - t0) / 1e3, (en - st) / 1e3, r.get("Queue_Id", "?")))
PY
done
find $OUT -name '*.csv' -size +8M -delete
