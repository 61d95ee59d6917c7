#!/bin/bash
# Two-rank rehearsal of bench.py's multi-rank path on a ONE-GPU box (both ranks on device 0, gloo for torch's barrier, the
# library's host transport for the gather because RCCL refuses two ranks on one device), then the full GPU suite.
# Usage: bash scripts/gpu_rehearse2.sh <tag>
set -u
TAG=${1:-rehearse2}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
( RAFTX_BENCH_BACKEND=gloo RAFTX_BENCH_DEVICE=0 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 \
    --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 10 --warmup 2 2>$OUT/bench2.err | tail -1 ) > $OUT/bench2.json
( timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -6 ) > $OUT/pytest_gpu.log
cut -c1-1500 $OUT/bench2.json; tail -3 $OUT/bench2.err; cat $OUT/pytest_gpu.log
