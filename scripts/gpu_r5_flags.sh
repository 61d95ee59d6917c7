#!/bin/bash
# bench.py flag combinations after the round-5 changes: every one must print a JSON line. Usage: bash scripts/gpu_r5_flags.sh <tag>
TAG=${1:-r05_flags}
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; cd $R
run() { name=$1; shift; ( timeout 400 "$@" 2>$OUT/$name.err | tail -1 ) > $OUT/$name.json; python - <<PY
import json
try:
    d=json.loads(open("$OUT/$name.json").read()); print("$name OK", round(d["value"]/1e6,1), d.get("ms_per_step"), d.get("n_gpus"), d.get("scaling"), d["config"].get("state"), d["config"].get("descriptors"))
except Exception as e:
    print("$name FAILED", e); print(open("$OUT/$name.err").read()[-600:])
PY
}
B="python bench.py --no-cpu-baseline --no-extra-legs --steps 4 --warmup 1"
run xi $B --xi-out
run nostream $B --no-stream
run resident $B --resident
run host_xi $B --descriptors host --xi-out
run host_nostream $B --descriptors host --no-stream
run depth3 $B --depth 3
run pageable $B --pageable --descriptors host
RAFTX_BENCH_DEVICE=0 run strong2 $B --gpus 2 --scaling strong --designs 4001
RAFTX_BENCH_DEVICE=0 run weak2 $B --gpus 2 --designs 2000
RAFTX_BENCH_DEVICE=0 run weak2_host $B --gpus 2 --designs 2000 --descriptors host
RAFTX_BENCH_DEVICE=0 run c4_2 python bench.py --gpus 2 --workload c4 --steps 2 --farms 50
RAFTX_BENCH_DEVICE=0 run c5_2 python bench.py --gpus 2 --workload c5 --steps 2 --sets 2
run profile python bench.py --steps 4 --warmup 1 --profile
run small python bench.py --no-cpu-baseline --no-extra-legs --steps 3 --warmup 1 --designs 7
