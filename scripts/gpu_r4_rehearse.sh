#!/bin/bash
# bench.py's own rank launcher on a ONE-GPU box: both ranks on device 0 (RCCL refuses that; the host transport carries the
# exchange steps and the JSON line says so).  C3 headline, C5 rows over ranks + SUM-reduce, C4 sea states over ranks.
set -u
TAG=${1:-r04_rehearse}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
( RAFTX_BENCH_DEVICE=0 timeout 600 python bench.py --gpus 2 --steps 10 --warmup 2 2>$OUT/c3.err | tail -1 ) > $OUT/c3.json
( RAFTX_BENCH_DEVICE=0 timeout 300 python bench.py --gpus 2 --workload c5 --steps 4 2>$OUT/c5.err | tail -1 ) > $OUT/c5.json
( RAFTX_BENCH_DEVICE=0 timeout 300 python bench.py --gpus 2 --workload c4 --steps 3 --farms 100 2>$OUT/c4.err | tail -1 ) > $OUT/c4.json
( timeout 100 python bench.py --gpus 2 2>&1 | tail -1 ) > $OUT/no_second_gpu.log
python - <<PY
import json
for n in ("c3", "c5", "c4"):
    try:
        d = json.load(open("$OUT/%s.json" % n))
        print(n, {k: d.get(k) for k in ("metric", "value", "n_gpus", "ms_per_step", "scaling", "per_rank_ms", "gather_ms", "scaling_efficiency", "single_rank_same_invocation")}, d["config"].get("gather"))
    except Exception as e:
        print(n, "FAILED", e, open("$OUT/%s.json" % n).read()[-300:], open("$OUT/%s.err" % n).read()[-1500:])
print(open("$OUT/no_second_gpu.log").read())
PY
