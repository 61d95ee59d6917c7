#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
timeout 900 python -m pytest tests/test_geometry.py -x -q -m gpu -k "crossing or soak or streamed" 2>&1 | tail -5
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --legs xi 2>/dev/null | tail -1 > gpurun_out/slab_bench.json
python - <<'PY'
import json
d = json.load(open("gpurun_out/slab_bench.json"))
print(d["ms_per_step"], d["value"]); print(json.dumps(d.get("xi_out"), indent=0)[:1500])
PY
