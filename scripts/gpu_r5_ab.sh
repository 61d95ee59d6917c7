#!/bin/bash
# parity suite + the bench line with device-made descriptors (default) and, on the same box, the host-descriptor form. Usage: bash scripts/gpu_r5_ab.sh <tag>
TAG=${1:-r05_ab}
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; cd $R
cat /sys/fs/cgroup/cpu.max > $OUT/host.txt 2>&1
( timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -25 ) > $OUT/pytest_gpu.log
( time timeout 900 python bench.py 2>$OUT/bench.err | tail -1 ) > $OUT/bench.json 2> $OUT/bench.time
( timeout 600 python bench.py --descriptors host --no-cpu-baseline --no-extra-legs 2>$OUT/bench_host.err | tail -1 ) > $OUT/bench_host.json
( timeout 600 python bench.py --no-cpu-baseline --no-extra-legs --steps 40 --warmup 5 2>$OUT/bench_dev40.err | tail -1 ) > $OUT/bench_dev40.json
cat $OUT/host.txt $OUT/pytest_gpu.log; tail -5 $OUT/bench.err $OUT/bench_host.err; cat $OUT/bench.time; python - <<PY
import json
for f in ("bench.json","bench_host.json","bench_dev40.json"):
    try:
        d=json.loads(open("$OUT/"+f).read())
        print(f, {k:d[k] for k in ("value","ms_per_step")}, d["roofline"]["frac"], d["roofline"]["kernel_ms_per_step"], d["step_breakdown_ms"], d["geometry"].get("host_params_ms_per_step"), d["geometry"]["host_descriptor_ms"], d.get("isolated_call",{}).get("ms_per_step"))
    except Exception as e:
        print(f, "ERR", e)
d=json.loads(open("$OUT/bench.json").read())
cb=d.get("cpu_baseline",{}); print({k:v for k,v in cb.items() if k not in ("port_simd","details","sample","implementation")}); print(cb.get("port_simd",{}).get("value"), cb.get("port_simd",{}).get("cores"))
print(d["parity"])
PY
