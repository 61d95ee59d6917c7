#!/bin/bash
# xi-out through bench.py's MAIN loop (--xi-out: staged four-deep), download stream low / high, 30 steps; one run with the
# per-step collection times.
set -u
TAG=${1:-r04_xi4}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
for P in low high; do
  ( RAFTX_D2H_PRIORITY=$P timeout 300 python bench.py --xi-out --steps 30 --warmup 3 --no-cpu-baseline --no-extra-legs 2>$OUT/main_$P.err | tail -1 ) > $OUT/main_$P.json
  python - <<PY
import json
d = json.load(open("$OUT/main_$P.json"))
print("$P main loop: ms/step %.3f kernel %.3f isolated %s" % (d["ms_per_step"], d["roofline"]["kernel_ms_per_step"], d.get("isolated_call", {}).get("ms_per_step")))
PY
done
( RAFTX_BENCH_DEBUG=1 RAFTX_D2H_PRIORITY=high timeout 300 python bench.py --xi-out --steps 12 --warmup 3 --no-cpu-baseline --no-extra-legs 2>&1 | grep collected | tail -12 )
