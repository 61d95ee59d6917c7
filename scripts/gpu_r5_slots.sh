#!/bin/bash
# XiLast scratch indexed by workgroup slot (xl_slot_acquire): parity of the fused-kernel tests, step / kernel time, and the
# FETCH_SIZE / WRITE_SIZE passes of whole-batch launches.  Usage: bash scripts/gpu_r5_slots.sh <tag>
set -u
TAG=${1:-r05_slots}
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; cd $R
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_hip_parity.py tests/test_hip_shapes.py tests/test_full_size.py tests/test_geometry.py -m gpu -x -q 2>&1 | tail -6 ) > $OUT/pytest.log
cat $OUT/pytest.log
( timeout 600 python bench.py --no-cpu-baseline --no-extra-legs --steps 40 --warmup 5 2>$OUT/bench.err | tail -1 ) > $OUT/bench.json
BENCH="python $R/bench.py --steps 5 --warmup 1 --profile"
cd /tmp
timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o bench -- $BENCH > $OUT/pmc_fetch.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o bench -- $BENCH > $OUT/pmc_write.log 2>&1
timeout 600 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/pmc_tcc -o bench -- $BENCH > $OUT/pmc_tcc.log 2>&1
cd $R
find $OUT -name '*.csv' -size +8M -delete
python - <<PY
import json, csv, glob, collections
d=json.loads(open("$OUT/bench.json").read())
print({k:d[k] for k in ("value","ms_per_step")}, d["roofline"]["frac"], d["roofline"]["kernel_ms_per_step"])
for name in ("pmc_fetch","pmc_write","pmc_tcc"):
    for f in glob.glob("$OUT/%s/**/*counter_collection.csv" % name, recursive=True):
        acc=collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if "k_solve_dynamics" in r["Kernel_Name"]:
                acc[(r["Counter_Name"], int(r["Grid_Size"]))].append(float(r["Counter_Value"]))
        for k,v in sorted(acc.items()): print(name, k, len(v), sum(v)/len(v))
PY
