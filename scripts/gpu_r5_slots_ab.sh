#!/bin/bash
# Same-box A/B: XiLast scratch per workgroup slot (default) against the per-pair slab (tuning build -DRAFTX_XL_PER_PAIR), alternating.
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
for rep in 1 2 3; do
  for v in "" _v_pair; do
    RAFTX_HIP_LIB=$R/raft_amd/csrc/libraftx_hip$v.so timeout 300 python bench.py --no-cpu-baseline --no-extra-legs --steps 40 --warmup 5 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('lib$v', round(d['ms_per_step'],4), round(d['roofline']['kernel_ms_per_step'],4), round(d['value']/1e6,1))"
  done
done
