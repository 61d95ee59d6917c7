#!/bin/bash
# k_geom_expand on its own high-priority stream (default) against the preparation stream, and host-made descriptors, same box, two repeats.
TAG=${1:-r05_prio}
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; cd $R
( timeout 300 python -m pytest tests -m gpu -x -q -k "variant" 2>&1 | tail -3 ) > $OUT/pytest.log
for rep in 1 2; do
  ( RAFTX_EXPAND_STREAM=1 timeout 300 python bench.py --no-cpu-baseline --no-extra-legs --steps 40 --warmup 5 2>>$OUT/bench.err | tail -1 ) > $OUT/b_prio_$rep.json
  ( RAFTX_EXPAND_STREAM=0 timeout 300 python bench.py --no-cpu-baseline --no-extra-legs --steps 40 --warmup 5 2>>$OUT/bench.err | tail -1 ) > $OUT/b_prep_$rep.json
  ( timeout 300 python bench.py --no-cpu-baseline --no-extra-legs --steps 40 --warmup 5 --descriptors host 2>>$OUT/bench.err | tail -1 ) > $OUT/b_host_$rep.json
done
cat $OUT/pytest.log; tail -3 $OUT/bench.err
python - <<PY
import json
for rep in (1,2):
    for m in ("prio","prep","host"):
        d=json.loads(open("$OUT/b_%s_%d.json"%(m,rep)).read()); print(m, rep, round(d["value"]/1e6,1), round(d["ms_per_step"],3), round(d["roofline"]["frac"],4), round(d["roofline"]["kernel_ms_per_step"],3))
PY
