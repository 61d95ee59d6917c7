#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(pwd)}
( timeout 600 python -m pytest tests/test_geometry.py tests/test_full_size.py -m gpu -x -q 2>&1 | tail -2 )
( timeout 200 python scripts/iso_xi.py 2>&1 | tail -4 | awk '{print $4}' | tr '\n' ' '; echo " <- default schedule" )
( RAFTX_BENCH_XI_STEPS=30 timeout 300 python bench.py --no-cpu-baseline --legs xi 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('bench: value %.1f M, xi_out %s' % (d['value']/1e6, {k: round(v,3) for k,v in d['xi_out'].items() if 'ms' in k}))" )
