#!/bin/bash
# Round 6: the generating form again, now that geom_design_block stages its descriptors in LDS.
TAG=${1:-r06_fused2}
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; cd $R
run() {  # name, env...
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
for rep in 1 2; do
  run base_$rep A=1
  run fused_$rep RAFTX_FUSED_GEN=1
  run base_nowait_$rep RAFTX_NO_MEMBER_WAIT=1
  run base_d2_$rep A=1 -- --depth 2
  run base_nowait_d2_$rep RAFTX_NO_MEMBER_WAIT=1 -- --depth 2
done 2>&1 | tee $OUT/ab.txt
