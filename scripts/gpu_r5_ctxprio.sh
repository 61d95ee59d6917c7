#!/bin/bash
TAG=${1:-r05_ctxprio}
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; cd $R
B="python bench.py --no-cpu-baseline --no-extra-legs --steps 40 --warmup 5"
for rep in 1 2; do
  ( RAFTX_CTX_PRIORITY=high timeout 300 $B 2>>$OUT/bench.err | tail -1 ) > $OUT/b_high_d3_$rep.json
  ( timeout 300 $B 2>>$OUT/bench.err | tail -1 ) > $OUT/b_norm_d3_$rep.json
  ( RAFTX_CTX_PRIORITY=high timeout 300 $B --depth 2 2>>$OUT/bench.err | tail -1 ) > $OUT/b_high_d2_$rep.json
  ( timeout 300 $B --depth 2 2>>$OUT/bench.err | tail -1 ) > $OUT/b_norm_d2_$rep.json
  ( RAFTX_CTX_PRIORITY=high timeout 300 $B --descriptors host 2>>$OUT/bench.err | tail -1 ) > $OUT/b_high_host_$rep.json
  ( timeout 300 $B --descriptors host 2>>$OUT/bench.err | tail -1 ) > $OUT/b_norm_host_$rep.json
done
tail -3 $OUT/bench.err
python - <<PY
import json, glob
for f in sorted(glob.glob("$OUT/b_*.json")):
    d=json.loads(open(f).read()); print(f.split("/")[-1], round(d["value"]/1e6,1), round(d["ms_per_step"],3), round(d["roofline"]["frac"],4), round(d["roofline"]["kernel_ms_per_step"],3))
PY
