#!/bin/bash
# Round 6: (1) phase split of k_geom_design with staged descriptors (timing build), (2) the 1 250-design shard: host time per
# staged call (RAFTX_SWEEP_DEBUG), one and two compute streams, depth 3 / 4, kernel timeline.
TAG=${1:-r06_shard}
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; cd $R
V=$R/raft_amd/csrc
RAFTX_HIP_LIB=$V/libraftx_hip_timing.so python scripts/bench_geom.py > $OUT/geom_timing.json 2> $OUT/geom_timing.err
grep -i "phases" $OUT/geom_timing.err | tail -2
sh1() {  # name, env... [-- args]
  local name=$1; shift
  local envs=() extra=()
  while [ $# -gt 0 ] && [ "$1" != "--" ]; do envs+=("$1"); shift; done
  [ $# -gt 0 ] && shift
  extra=("$@")
  ( env "${envs[@]}" timeout 300 python bench.py --no-cpu-baseline --designs 1250 --no-extra-legs --steps 200 --warmup 20 "${extra[@]}" 2>$OUT/sh_$name.err | tail -1 ) > $OUT/sh_$name.json
  python - $OUT/sh_$name.json $name <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1])); r = d["roofline"]
    print("%-28s step %.4f ms  kernel(union) %.4f  per-launch %.4f  gen_sum %.3f stats %.3f" % (sys.argv[2], d["ms_per_step"], r["kernel_ms_per_step"], r.get("kernel_ms_per_launch", 0.0),
          d["step_breakdown_ms"]["generation_kernels_sum"], d["step_breakdown_ms"]["statistics_kernels_sum"]), flush=True)
except Exception as e:
    print(sys.argv[2], "FAILED", e, flush=True)
PY
}
for rep in 1 2; do
  sh1 d3_$rep A=1
  sh1 d3_2s_$rep RAFTX_SWEEP_STREAMS=2
  sh1 d4_$rep A=1 -- --depth 4
  sh1 d4_2s_$rep RAFTX_SWEEP_STREAMS=2 -- --depth 4
  sh1 d2_$rep A=1 -- --depth 2
done 2>&1 | tee $OUT/ab.txt
( RAFTX_SWEEP_DEBUG=1 timeout 300 python bench.py --no-cpu-baseline --designs 1250 --no-extra-legs --steps 30 --warmup 10 2>&1 | grep "raftx_sweep slot" | tail -12 ) > $OUT/host_debug.txt
cat $OUT/host_debug.txt
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --hip-trace --output-format csv -d $OUT/trace -o bench -- python $R/bench.py --designs 1250 --steps 30 --warmup 10 --profile --no-cpu-baseline --no-extra-legs > $OUT/trace.log 2>&1
cd $R
python - $OUT <<'PY' | tee $OUT/timelines.txt
import csv, sys, os
p = os.path.join(sys.argv[1], "trace", "bench_kernel_trace.csv")
rows = list(csv.DictReader(open(p)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
fused = [i for i, r in enumerate(rows) if r["Kernel_Name"].startswith("raftx_kp")]
print("fused launches", len(fused))
i0 = fused[25]; t0 = int(rows[i0]["Start_Timestamp"])
for r in rows[i0 - 4:i0 + 30]:
    s = (int(r["Start_Timestamp"]) - t0) / 1e3; e = (int(r["End_Timestamp"]) - t0) / 1e3
    print("%-30s start %9.1f end %9.1f dur %8.1f q%s grid %s" % (r["Kernel_Name"][:30], s, e, e - s, r["Queue_Id"], r["Grid_Size_X"]))
# host API time per step
p = os.path.join(sys.argv[1], "trace", "bench_hip_api_trace.csv")
if os.path.exists(p):
    api = list(csv.DictReader(open(p)))
    from collections import defaultdict
    tot = defaultdict(lambda: [0, 0])
    for a in api:
        n = a.get("Function") or a.get("Name")
        tot[n][0] += 1; tot[n][1] += int(a["End_Timestamp"]) - int(a["Start_Timestamp"])
    for n, (c, t) in sorted(tot.items(), key=lambda kv: -kv[1][1])[:14]:
        print("%-40s calls %7d total %9.1f ms  avg %7.2f us" % (n, c, t / 1e6, t / 1e3 / c))
PY
find $OUT -name '*_trace.csv' -size +6M -delete
