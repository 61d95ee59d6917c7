#!/bin/bash
# streamed step: depth 2 / 3 / 4 with device-made and host-made descriptors, same box. Usage: bash scripts/gpu_r5_depth.sh <tag>
TAG=${1:-r05_depth}
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; cd $R
for mode in device host; do for depth in 2 3 4; do
  ( timeout 300 python bench.py --no-cpu-baseline --no-extra-legs --steps 40 --warmup 5 --descriptors $mode --depth $depth 2>>$OUT/bench.err | tail -1 ) > $OUT/b_${mode}_$depth.json
done; done
tail -3 $OUT/bench.err
python - <<PY
import json
for mode in ("device","host"):
    for depth in (2,3,4):
        d=json.loads(open("$OUT/b_%s_%d.json"%(mode,depth)).read()); print(mode, depth, round(d["value"]/1e6,1), round(d["ms_per_step"],3), round(d["roofline"]["frac"],4), round(d["roofline"]["kernel_ms_per_step"],3), d["step_breakdown_ms"])
PY
