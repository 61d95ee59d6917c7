#!/bin/bash
# The driver's N-rank command line on a one-GPU box (all ranks on device 0, host transport): N = 2, 4, 8, weak and strong.
TAG=${1:-r05_rehearse}
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; cd $R
for n in 2 4 8; do
  ( RAFTX_BENCH_DEVICE=0 timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port $((29500+n)) bench.py --gpus $n --steps 5 --warmup 2 --designs 2500 2>$OUT/weak$n.err | tail -1 ) > $OUT/weak$n.json
  ( RAFTX_BENCH_DEVICE=0 timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port $((29600+n)) bench.py --gpus $n --steps 5 --warmup 2 --scaling strong 2>$OUT/strong$n.err | tail -1 ) > $OUT/strong$n.json
done
python - <<PY
import json
for n in (2,4,8):
    for m in ("weak","strong"):
        try:
            d=json.load(open("$OUT/%s%d.json"%(m,n))); print(m, n, round(d["value"]/1e6,1), round(d["ms_per_step"],3), d["scaling"], d["config"]["shard_designs"], d["config"]["gather"][:40], d["parity"]["niter_mismatches_vs_reference"], len(d["host_descriptor_ms_per_rank"]["all"]))
        except Exception as e:
            print(m, n, "FAILED", e); print(open("$OUT/%s%d.err"%(m,n)).read()[-500:])
PY
