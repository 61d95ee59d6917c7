#!/bin/bash
# the default-length run (K = 10, W = 2) and K = 20 several times on one box: what the driver's single run sees
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/${1:-r06_k10}; mkdir -p $OUT; cd $R
for rep in 1 2 3; do
  for k in 10 20 40; do
    for mw in 0 1; do
      ( RAFTX_MEMBER_WAIT=$mw timeout 300 python bench.py --no-cpu-baseline --no-extra-legs --steps $k --warmup 2 2>/dev/null | tail -1 ) > $OUT/b.json
      python - $OUT/b.json $k $mw <<'PY'
import json, sys
d = json.load(open(sys.argv[1])); r = d["roofline"]
print("K=%s member_wait=%s step %.4f kernel(union) %.4f per-launch %.4f frac %.4f value %.1f M" % (sys.argv[2], sys.argv[3], d["ms_per_step"], r["kernel_ms_per_step"], r["kernel_ms_per_launch"], r["frac"], d["value"] / 1e6), flush=True)
PY
    done
  done
done | tee $OUT/k.txt
