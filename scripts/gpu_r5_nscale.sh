#!/bin/bash
# Fused kernel time against the number of pairs of a launch (same designs drawn from one stream): intercept = what a launch costs beyond its pairs.
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
cat > /tmp/_row.py <<'PY'
import json, sys
d = json.loads(sys.stdin.read())
k = d['kernel_resident']
print(sys.argv[1], 'kernel_ms', round(k['kernel_ms'], 4), 'us_per_pair', round(1e3 * k['kernel_ms'] / int(sys.argv[1]), 4), 'mean_it', round(d['mean_iterations'], 3))
PY
for n in 1024 2048 4096 8192 10000 10240 16384 20000 32768 40000; do
  timeout 300 python bench.py --no-cpu-baseline --no-extra-legs --designs $n --steps 6 --warmup 2 --resident 2>/dev/null | tail -1 | python /tmp/_row.py $n
done
