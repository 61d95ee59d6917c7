#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
for ARGS in "--no-cpu-baseline --legs xi" "--steps 4 --warmup 1 --no-cpu-baseline --legs xi"; do
echo "== $ARGS"
( RAFTX_BENCH_DEBUG=1 timeout 300 python bench.py $ARGS 2>&1 | grep "xi step" | awk '{print $6}' | python -c "
import sys
t=[float(x) for x in sys.stdin.read().split()]
print(' '.join('%.1f'%(b-a) for a,b in zip(t,t[1:])))" )
done
