#!/bin/bash
# Round 6: the gap between two fused kernels of a streamed sweep.  A/B on one box: k_geom_design with its descriptors staged
# in LDS (default build) against the previous form (libraftx_hip_v_nostage.so), the generation stream in the highest
# priority class, the fused kernel not waiting for the member passes queued behind it, the scan as a 256-thread block.
TAG=${1:-r06_gap}
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; cd $R
V=$R/raft_amd/csrc
( timeout 900 python -m pytest tests/test_geometry.py -m gpu -x -q 2>&1 | tail -8 ) > $OUT/pytest_geom.log
cat $OUT/pytest_geom.log
run() {  # name, env...
  local name=$1; shift
  ( env "$@" timeout 300 python bench.py --no-cpu-baseline --no-extra-legs --steps 40 --warmup 10 2>$OUT/bench_$name.err | tail -1 ) > $OUT/bench_$name.json
  python - $OUT/bench_$name.json $name <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    r = d["roofline"]
    print("%-34s step %.4f ms  kernel(union) %.4f  per-launch %.4f  frac %.4f  step_frac %.4f  value %.1f M  gen %.3f" % (sys.argv[2], d["ms_per_step"], r["kernel_ms_per_step"],
          r.get("kernel_ms_per_launch", 0.0), r["frac"], r.get("step_frac", 0.0), d["value"] / 1e6, d["step_breakdown_ms"]["generation_kernels_sum"]), flush=True)
except Exception as e:
    print(sys.argv[2], "FAILED", e, flush=True)
PY
}
NS=RAFTX_HIP_LIB=$V/libraftx_hip_v_nostage.so
for rep in 1 2; do
  run nostage_$rep $NS
  run stage_$rep A=1
  run stage_prio_$rep RAFTX_GEN_PRIORITY=high
  run stage_nowait_$rep RAFTX_NO_MEMBER_WAIT=1
  run stage_nowait_scan256_$rep RAFTX_NO_MEMBER_WAIT=1 RAFTX_SCAN_T=256
  run stage_prio_nowait_scan256_$rep RAFTX_GEN_PRIORITY=high RAFTX_NO_MEMBER_WAIT=1 RAFTX_SCAN_T=256
  run nostage_nowait_scan256_$rep $NS RAFTX_NO_MEMBER_WAIT=1 RAFTX_SCAN_T=256
done 2>&1 | tee $OUT/ab.txt
( env RAFTX_NO_MEMBER_WAIT=1 RAFTX_SCAN_T=256 timeout 300 python bench.py --no-cpu-baseline --no-extra-legs --steps 40 --warmup 10 --depth 4 2>/dev/null | tail -1 ) > $OUT/bench_d4.json
python - $OUT/bench_d4.json <<'PY' | tee -a $OUT/ab.txt
import json, sys
d = json.load(open(sys.argv[1])); r = d["roofline"]
print("stage_nowait_scan256_depth4 step %.4f kernel %.4f frac %.4f" % (d["ms_per_step"], r["kernel_ms_per_step"], r["frac"]))
PY
cd /tmp; export TMPDIR=/tmp
trace() {  # name, env...
  local name=$1; shift
  env "$@" timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_$name -o bench -- python $R/bench.py --steps 20 --warmup 10 --profile --no-cpu-baseline --no-extra-legs > $OUT/trace_$name.log 2>&1
}
trace stage A=1
trace nowait RAFTX_NO_MEMBER_WAIT=1 RAFTX_SCAN_T=256
trace prio_nowait RAFTX_GEN_PRIORITY=high RAFTX_NO_MEMBER_WAIT=1 RAFTX_SCAN_T=256
cd $R
python - $OUT <<'PY' | tee $OUT/timelines.txt
import csv, sys, os
for v in ("stage", "nowait", "prio_nowait"):
    p = os.path.join(sys.argv[1], "trace_" + v, "bench_kernel_trace.csv")
    if not os.path.exists(p):
        print("no trace", v); continue
    rows = list(csv.DictReader(open(p)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    fused = [i for i, r in enumerate(rows) if r["Kernel_Name"].startswith("raftx_kp")]
    print("===", v, "fused launches", len(fused))
    if len(fused) < 24: continue
    i0 = fused[18]; t0 = int(rows[i0]["Start_Timestamp"])
    for r in rows[i0 - 7:i0 + 24]:
        s = (int(r["Start_Timestamp"]) - t0) / 1e3; e = (int(r["End_Timestamp"]) - t0) / 1e3
        print("%-30s start %9.1f end %9.1f dur %8.1f q%s grid %s" % (r["Kernel_Name"][:30], s, e, e - s, r["Queue_Id"], r["Grid_Size_X"]))
PY
find $OUT -name '*_kernel_trace.csv' -size +4M -delete
for v in stage nowait prio_nowait; do cut -c1-150 $OUT/trace_$v/bench_kernel_stats.csv | head -9; done
