#!/bin/bash
# QTF kernel visit: parity (tests/test_hip_qtf.py, C5 full size), timing with 128- and 64-thread rows, xi-out legs.
set -u
TAG=${1:-r04_qtf}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
( timeout 600 python -m pytest tests/test_hip_qtf.py tests/test_full_size.py tests/test_geometry.py -m gpu -x -q 2>&1 | tail -15 ) > $OUT/pytest.log
( timeout 300 python scripts/bench_qtf.py 16 2>&1 | tail -3 ) > $OUT/qtf_128.jsonl
( RAFTX_QTF_BLOCK=64 timeout 300 python scripts/bench_qtf.py 16 2>&1 | tail -3 ) > $OUT/qtf_64.jsonl
( timeout 600 python bench.py --no-cpu-baseline 2>$OUT/bench.err | tail -1 ) > $OUT/bench.json
cat $OUT/pytest.log; cat $OUT/qtf_128.jsonl $OUT/qtf_64.jsonl | cut -c1-400
python - <<PY
import json
d = json.load(open("$OUT/bench.json"))
print("value %.1f M ms/step %.3f kernel %.3f" % (d["value"]/1e6, d["ms_per_step"], d["roofline"]["kernel_ms_per_step"]))
print("xi_out", d.get("xi_out")); print("isolated", d.get("isolated_call"))
print("c5", json.dumps(d.get("c5_qtf"))[:1200])
PY
tail -5 $OUT/bench.err
