#!/bin/bash
# long streaks: 2 000 streamed crossings (stats out), 300 with the responses downloaded; device memory before / after
cd ${GRAFT_REPO_ROOT:-$(pwd)}
rocm-smi --showmemuse 2>/dev/null | grep -i "GPU\[0\].*VRAM" | head -2
( timeout 600 python bench.py --no-cpu-baseline --no-extra-legs --steps 2000 --warmup 5 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('2000 steps: ms/step %.3f value %.1f M kernel %.3f parity %s' % (d['ms_per_step'], d['value']/1e6, d['roofline']['kernel_ms_per_step'], d['parity']))" )
( timeout 600 python bench.py --no-cpu-baseline --no-extra-legs --xi-out --steps 300 --warmup 5 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('300 xi-out steps: ms/step %.3f value %.1f M' % (d['ms_per_step'], d['value']/1e6))" )
rocm-smi --showmemuse 2>/dev/null | grep -i "GPU\[0\].*VRAM" | head -2
