#!/bin/bash
# Round 4, first visit: parity suite, default bench (with the C2/C4/C5 legs), two-rank rehearsals of bench.py's own launcher
# on one GPU (host transport), the sharded C4 / C5 workloads.
set -u
TAG=${1:-r04_first}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
( timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 ) > $OUT/pytest_gpu.log
( timeout 600 python bench.py 2>$OUT/bench.err | tail -1 ) > $OUT/bench.json
( RAFTX_BENCH_DEVICE=0 timeout 600 python bench.py --gpus 2 --steps 10 --warmup 2 2>$OUT/bench2.err | tail -1 ) > $OUT/bench2.json
( RAFTX_BENCH_DEVICE=0 timeout 300 python bench.py --gpus 2 --workload c5 --steps 4 2>$OUT/bench2_c5.err | tail -1 ) > $OUT/bench2_c5.json
( RAFTX_BENCH_DEVICE=0 timeout 300 python bench.py --gpus 2 --workload c4 --steps 3 2>$OUT/bench2_c4.err | tail -1 ) > $OUT/bench2_c4.json
( timeout 300 python bench.py --gpus 2 2>&1 | tail -2 ) > $OUT/bench_gpus2_on_one_gpu.log
cat $OUT/pytest_gpu.log
for f in bench bench2 bench2_c5 bench2_c4; do echo "== $f"; tail -3 $OUT/$f.err; head -c 600 $OUT/$f.json; echo; done
cat $OUT/bench_gpus2_on_one_gpu.log
