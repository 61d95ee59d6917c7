#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
cat > /tmp/_row.py <<'PY'
import json, sys
d = json.loads(sys.stdin.read())
k = d['kernel_resident']
print(sys.argv[1], 'resident kernel_ms', round(k['kernel_ms'], 4), 'stream ms/step', round(d['ms_per_step'], 4), 'kernel/step', round(d['roofline']['kernel_ms_per_step'], 4), round(d['value'] / 1e6, 1))
PY
rocm-smi --showperflevel --showclocks 2>&1 | grep -v "^$" | head -20
run() { timeout 300 python bench.py --no-cpu-baseline --no-extra-legs --steps 30 --warmup 3 --resident 2>/dev/null | tail -1 | python /tmp/_row.py "$1"; }
run auto
rocm-smi --setperflevel high 2>&1 | tail -3
rocm-smi --showperflevel 2>&1 | grep -i "perf" | head -3
run high
run high
rocm-smi --setperfdeterminism 2400 2>&1 | tail -3
run determinism2400
rocm-smi --resetperfdeterminism 2>&1 | tail -2
rocm-smi --setperflevel auto 2>&1 | tail -2
run auto_again
