#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
( timeout 600 python -m pytest tests/test_flexible.py tests/test_hip_parity.py -m gpu -x -q 2>&1 | tail -4 )
( timeout 300 python scripts/bench_dropin.py --repeat 5 2>&1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
for c in d['configs']:
    print(c['config'][:40], [round(x['gpu_call_ms_median'],3) for x in c['cases']], [x['rel_err_vs_reference'] for x in c['cases']][:2])" )
( RAFTX_DENSE_L2=1 timeout 300 python scripts/bench_dropin.py --repeat 5 2>&1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
c=d['configs'][-1]; print('L2-workspace kernel:', [round(x['gpu_call_ms_median'],3) for x in c['cases']])" )
