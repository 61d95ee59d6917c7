#!/bin/bash
# Two-rank rehearsal of bench.py's multi-rank path on a ONE-GPU box (both ranks on device 0; the library's host transport
# carries barrier / max / gather because RCCL refuses two ranks on one device -- the JSON line says so).
# Usage: bash scripts/gpu_rehearse2.sh <tag>
set -u
TAG=${1:-rehearse2}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
( RAFTX_BENCH_DEVICE=0 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 \
    --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 10 --warmup 2 2>$OUT/bench2.err | tail -1 ) > $OUT/bench2.json
python - <<PY
import json
d = json.load(open("$OUT/bench2.json"))
print({k: d[k] for k in ("value", "n_gpus", "ms_per_step")}, d["config"]["gather"], d["roofline"]["frac"])
PY
tail -3 $OUT/bench2.err
