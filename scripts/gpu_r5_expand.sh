#!/bin/bash
# k_geom_expand after the coalescing change: variant tests + kernel trace of a short streamed run. Usage: bash scripts/gpu_r5_expand.sh <tag>
TAG=${1:-r05_expand}
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; cd $R
( timeout 600 python -m pytest tests -m gpu -x -q -k "variant" 2>&1 | tail -4 ) > $OUT/pytest.log
( timeout 600 python bench.py --no-cpu-baseline --no-extra-legs --steps 40 --warmup 5 2>$OUT/bench.err | tail -1 ) > $OUT/bench40.json
( timeout 600 python bench.py --no-cpu-baseline --no-extra-legs --steps 40 --warmup 5 --descriptors host 2>>$OUT/bench.err | tail -1 ) > $OUT/bench40_host.json
export TMPDIR=/tmp; cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o bench -- python $R/bench.py --steps 20 --warmup 5 --profile > $OUT/trace.log 2>&1
cd $R; find $OUT -name '*_kernel_trace.csv' -size +4M -delete
cat $OUT/pytest.log; tail -3 $OUT/bench.err; cut -c1-120 $OUT/trace/bench_kernel_stats.csv | head -6
python - <<PY
import json
for f in ("bench40.json","bench40_host.json"):
    d=json.loads(open("$OUT/"+f).read()); print(f, d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["kernel_ms_per_step"], d["step_breakdown_ms"])
PY
