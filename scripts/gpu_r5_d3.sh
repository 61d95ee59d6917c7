#!/bin/bash
TAG=${1:-r05_d3}
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; cd $R
( timeout 600 python -m pytest tests/test_bench_launcher.py -m gpu -x -q 2>&1 | tail -3 ) > $OUT/pytest.log
for rep in 1 2; do
( timeout 300 python bench.py --no-cpu-baseline --no-extra-legs 2>>$OUT/bench.err | tail -1 ) > $OUT/b_default_$rep.json
( timeout 300 python bench.py --no-cpu-baseline --no-extra-legs --depth 2 2>>$OUT/bench.err | tail -1 ) > $OUT/b_depth2_$rep.json
( timeout 300 python bench.py --no-cpu-baseline --no-extra-legs --steps 20 --warmup 5 2>>$OUT/bench.err | tail -1 ) > $OUT/b_default20_$rep.json
( timeout 300 python bench.py --no-cpu-baseline --no-extra-legs --steps 20 --warmup 5 --depth 2 2>>$OUT/bench.err | tail -1 ) > $OUT/b_depth2_20_$rep.json
( timeout 300 python bench.py --no-cpu-baseline --no-extra-legs --steps 20 --warmup 5 --descriptors host 2>>$OUT/bench.err | tail -1 ) > $OUT/b_host20_$rep.json
done
cat $OUT/pytest.log; tail -3 $OUT/bench.err
python - <<PY
import json, glob
for f in sorted(glob.glob("$OUT/b_*.json")):
    d=json.loads(open(f).read()); print(f.split("/")[-1], round(d["value"]/1e6,1), round(d["ms_per_step"],3), round(d["roofline"]["frac"],4), round(d["roofline"]["kernel_ms_per_step"],3))
PY
