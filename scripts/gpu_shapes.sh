#!/bin/bash
# bench the lean sweep kernel under several launch shapes. Usage: bash scripts/gpu_shapes.sh <tag> "4,64 2,128 1,256"
TAG=${1:-s}; SHAPES=${2:-"4,64 2,128 1,256"}
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; cd $R
( timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 ) > $OUT/pytest_gpu.log
cat $OUT/pytest_gpu.log
for sh in $SHAPES; do
  echo "== shape $sh"
  RAFTX_SHAPE=$sh timeout 300 python bench.py --no-cpu-baseline --steps 5 --warmup 1 2>&1 | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print("value %.4g dcf/s  kernel_ms %.3f  err %.2e" % (d["value"], d["roofline"]["kernel_ms"], d["rao_max_rel_err_vs_reference"]))' 2>&1 | tee -a $OUT/shapes.log
done
