#!/bin/bash
# stream-priority A/B of the streamed bench. Usage: bash scripts/gpu_prio.sh <tag>
TAG=${1:-prio}
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; cd $R
run() { echo "== $1" >> $OUT/log; ( env $2 timeout 300 python bench.py --no-cpu-baseline --steps 30 --warmup 5 2>&1 | tail -1 | python -c '
import sys, json
d = json.loads(sys.stdin.read())
print("value %.1f M  ms/step %.3f  %s  isolated %.3f" % (d["value"] / 1e6, d["ms_per_step"], d["step_breakdown_ms"], d["isolated_call"]["ms_per_step"]))' ) >> $OUT/log 2>&1; }
run "priority (default)" "X=1"
run "flat" "RAFTX_SWEEP_FLAT_PRIORITY=1"
run "priority + generation overlap" "RAFTX_SWEEP_GEN_OVERLAP=1"
run "priority (default) again" "X=1"
cat $OUT/log
