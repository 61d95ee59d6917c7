#!/bin/bash
# Round 6: the member pass with its station rows in registers (one batched load per station, everything stored at the end)
# against the previous form (libraftx_hip_v_oldmember.so): parity suite of the generator, kernels alone, the streamed step.
TAG=${1:-r06_member}
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; cd $R
V=$R/raft_amd/csrc
( timeout 900 python -m pytest tests/test_geometry.py tests/test_hip_parity.py -m gpu -x -q 2>&1 | tail -4 ) > $OUT/pytest_geom.log
cat $OUT/pytest_geom.log
run() {  # name, env... [-- bench args]
  local name=$1; shift
  local envs=() extra=()
  while [ $# -gt 0 ] && [ "$1" != "--" ]; do envs+=("$1"); shift; done
  [ $# -gt 0 ] && shift
  extra=("$@")
  ( env "${envs[@]}" timeout 300 python bench.py --no-cpu-baseline --no-extra-legs --steps 40 --warmup 10 "${extra[@]}" 2>$OUT/bench_$name.err | tail -1 ) > $OUT/bench_$name.json
  python - $OUT/bench_$name.json $name <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    r = d["roofline"]
    print("%-34s step %.4f ms  kernel(union) %.4f  per-launch %.4f  frac %.4f  step_frac %.4f  value %.1f M" % (sys.argv[2], d["ms_per_step"], r["kernel_ms_per_step"],
          r.get("kernel_ms_per_launch", 0.0), r["frac"], r.get("step_frac", 0.0), d["value"] / 1e6), flush=True)
except Exception as e:
    print(sys.argv[2], "FAILED", e, flush=True)
PY
}
OLD=RAFTX_HIP_LIB=$V/${RAFTX_AB_OLD:-libraftx_hip_v_oldmember.so}
for rep in 1 2 3; do
  run old_$rep $OLD
  run new_$rep A=1
done 2>&1 | tee $OUT/ab.txt
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_geom -o geom -- python $R/scripts/bench_geom.py > $OUT/trace_geom.log 2>&1
cut -c1-120 $OUT/trace_geom/geom_kernel_stats.csv | head -8
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_new -o bench -- python $R/bench.py --steps 20 --warmup 10 --profile --no-cpu-baseline --no-extra-legs > $OUT/trace_new.log 2>&1
cd $R
python - $OUT <<'PY' | tee $OUT/timelines.txt
import csv, sys, os
p = os.path.join(sys.argv[1], "trace_new", "bench_kernel_trace.csv")
rows = list(csv.DictReader(open(p)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
fused = [i for i, r in enumerate(rows) if r["Kernel_Name"].startswith("raftx_kp")]
i0 = fused[18]; t0 = int(rows[i0]["Start_Timestamp"])
for r in rows[i0 - 7:i0 + 12]:
    s = (int(r["Start_Timestamp"]) - t0) / 1e3; e = (int(r["End_Timestamp"]) - t0) / 1e3
    print("%-30s start %9.1f end %9.1f dur %8.1f q%s grid %s" % (r["Kernel_Name"][:30], s, e, e - s, r["Queue_Id"], r["Grid_Size_X"]))
PY
find $OUT -name '*_kernel_trace.csv' -size +4M -delete
