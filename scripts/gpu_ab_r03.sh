#!/bin/bash
# same-box A/B of the headline step: the tree of the previous round (built under build/r03 from `git archive`) against this one
F="--gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-extra-legs"
O=$PWD/gpurun_out/ab_r03
mkdir -p $O
for i in 1 2 3; do
  (cd build/r03 && python bench.py $F 2>/dev/null | tail -1) >> $O/r03.jsonl
  python bench.py $F 2>/dev/null | tail -1 >> $O/r04.jsonl
done
python - <<'PY'
import json
for t in ("r03", "r04"):
    for l in open("gpurun_out/ab_r03/%s.jsonl" % t):
        d = json.loads(l)
        print(t, "%.3f ms/step" % d["ms_per_step"], "kernel", d.get("kernel_ms_per_step"), "frac", d["roofline"].get("frac"))
PY
