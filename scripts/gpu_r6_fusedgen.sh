#!/bin/bash
# Round 6: the generating form of the fused kernel (raftx_kpg_f0, raftx_fusedgen.h) -- parity tests, then the headline step
# with and without it, with the statistics on their own stream, with alternating main streams, at depth 4; kernel timeline.
TAG=${1:-r06_fusedgen}
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; cd $R
( timeout 600 python -m pytest tests/test_geometry.py tests/test_code_object.py -m gpu -x -q -k "fused_generation or streamed or staged or crossing or code" 2>&1 | tail -15 ) > $OUT/pytest_fused.log
cat $OUT/pytest_fused.log
run() {  # name, env...
  local name=$1; shift
  ( env "$@" timeout 300 python bench.py --no-cpu-baseline --no-extra-legs --steps 40 --warmup 10 2>$OUT/bench_$name.err | tail -1 ) > $OUT/bench_$name.json
  python - $OUT/bench_$name.json $name <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    r = d["roofline"]
    print("%-28s step %.4f ms  kernel(union) %.4f  per-launch %.4f  frac %.4f  step_frac %.4f  value %.1f M" % (sys.argv[2], d["ms_per_step"], r["kernel_ms_per_step"],
          r.get("kernel_ms_per_launch", 0.0), r["frac"], r.get("step_frac", 0.0), d["value"] / 1e6), flush=True)
except Exception as e:
    print(sys.argv[2], "FAILED", e, flush=True)
PY
}
for rep in 1 2; do
  run base_$rep RAFTX_FUSED_GEN=0
  run fused_$rep RAFTX_FUSED_GEN=1
  run fused_stats_$rep RAFTX_FUSED_GEN=1 RAFTX_STATS_STREAM=1
  run fused_2s_$rep RAFTX_FUSED_GEN=1 RAFTX_SWEEP_STREAMS=2
  run fused_2s_stats_$rep RAFTX_FUSED_GEN=1 RAFTX_SWEEP_STREAMS=2 RAFTX_STATS_STREAM=1
  run fused_waitmem_$rep RAFTX_FUSED_GEN=1 RAFTX_FUSED_WAIT_MEMBER=1
done 2>&1 | tee $OUT/ab.txt
( env RAFTX_FUSED_GEN=1 RAFTX_STATS_STREAM=1 timeout 300 python bench.py --no-cpu-baseline --no-extra-legs --steps 40 --warmup 10 --depth 4 2>/dev/null | tail -1 ) > $OUT/bench_d4.json
python - $OUT/bench_d4.json <<'PY' | tee -a $OUT/ab.txt
import json, sys
d = json.load(open(sys.argv[1])); r = d["roofline"]
print("fused_stats_depth4 step %.4f kernel %.4f frac %.4f" % (d["ms_per_step"], r["kernel_ms_per_step"], r["frac"]))
PY
cd /tmp; export TMPDIR=/tmp
for v in 0 1; do
  RAFTX_FUSED_GEN=$v timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_$v -o bench -- python $R/bench.py --steps 20 --warmup 10 --profile --no-cpu-baseline --no-extra-legs > $OUT/trace_$v.log 2>&1
done
RAFTX_FUSED_GEN=1 RAFTX_STATS_STREAM=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_1s -o bench -- python $R/bench.py --steps 20 --warmup 10 --profile --no-cpu-baseline --no-extra-legs > $OUT/trace_1s.log 2>&1
cd $R
python - $OUT <<'PY' | tee $OUT/timelines.txt
import csv, sys, os
for v in ("0", "1", "1s"):
    p = os.path.join(sys.argv[1], "trace_" + v, "bench_kernel_trace.csv")
    if not os.path.exists(p):
        print("no trace", v); continue
    rows = list(csv.DictReader(open(p)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    fused = [i for i, r in enumerate(rows) if r["Kernel_Name"].startswith("raftx_kp")]
    print("=== RAFTX_FUSED_GEN /stats", v, "fused launches", len(fused))
    if len(fused) < 24: continue
    i0 = fused[18]; t0 = int(rows[i0]["Start_Timestamp"])
    for r in rows[i0 - 6:i0 + 30]:
        s = (int(r["Start_Timestamp"]) - t0) / 1e3; e = (int(r["End_Timestamp"]) - t0) / 1e3
        print("%-30s start %9.1f end %9.1f dur %8.1f q%s grid %s" % (r["Kernel_Name"][:30], s, e, e - s, r["Queue_Id"], r["Grid_Size_X"]))
PY
find $OUT -name '*_kernel_trace.csv' -size +4M -delete
for v in 0 1 1s; do cut -c1-150 $OUT/trace_$v/bench_kernel_stats.csv | head -8; done
