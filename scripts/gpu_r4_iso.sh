#!/bin/bash
# isolated xi-out call: slabs x generation overlap
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
for V in "RAFTX_XI_SLABS=5" "RAFTX_XI_SLABS=4" "RAFTX_XI_SLABS=3" "RAFTX_XI_SLABS=6" "RAFTX_XI_SLABS=8"; do
  ( export $V; RAFTX_BENCH_XI_STEPS=10 timeout 300 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --legs xi 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); x=d['xi_out']; print('$V', 'isolated %.3f streamed %.3f' % (x['isolated_ms_per_step'], x['streamed_ms_per_step']))" )
done
