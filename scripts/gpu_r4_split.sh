#!/bin/bash
# streamed step with the batch cut into two / three blocks (the first one small: its fused kernel hides the generation of the rest)
cd ${GRAFT_REPO_ROOT:-$(pwd)}
for V in "" "0.1,0.9" "0.15,0.85" "0.2,0.8" "0.1024,0.8976" "0.3,0.7" "0.1,0.2,0.7"; do
  ( [ -n "$V" ] && export RAFTX_SWEEP_SPLIT=$V; timeout 300 python bench.py --no-cpu-baseline --no-extra-legs --steps 40 --warmup 5 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('split [%s]: ms/step %.3f  kernel sum %.3f  value %.1f M' % ('$V', d['ms_per_step'], d['roofline']['kernel_ms_per_step'], d['value']/1e6))" )
done
