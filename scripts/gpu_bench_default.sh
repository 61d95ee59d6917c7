#!/bin/bash
# default bench.py run (what the driver runs) -> gpurun_out/<tag>/bench.json + a short summary
set -u
TAG=${1:-bench}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
( timeout 900 python bench.py "${@:2}" 2>$OUT/bench.err | tail -1 ) > $OUT/bench.json
python - <<PY
import json
d = json.load(open("$OUT/bench.json"))
print("value %.1f M  ms/step %.3f  kernel %.3f  frac %.3f" % (d["value"]/1e6, d["ms_per_step"], d["roofline"]["kernel_ms_per_step"], d["roofline"]["frac"]))
for k in ("isolated_call", "xi_out", "cpu_baseline"):
    print(k, json.dumps(d.get(k))[:400])
for k in ("c2_dropin", "c4_farm", "c5_qtf"):
    v = d.get(k)
    if v: print(k, json.dumps({a: b for a, b in v.items() if a not in ("cases", "dropin_potSecOrder1_calls", "note", "config", "golden")})[:900])
PY
tail -3 $OUT/bench.err
