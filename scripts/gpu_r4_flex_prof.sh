#!/bin/bash
# kernel-trace statistics and counters of the batched flexible sweep (scripts/bench_flex.py 16: 16 units x 3 sea states)
set -u
TAG=${1:-r04_flex}
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o flex -- python $R/scripts/bench_flex.py 16 > $OUT/trace.log 2>&1
tail -1 $OUT/trace.log
cp $(find $OUT/trace -name '*kernel_stats.csv' | head -1) $OUT/kernel_stats.csv
head -12 $OUT/kernel_stats.csv | cut -c1-200
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU --kernel-trace --output-format csv -d $OUT/pmc -o flex -- python $R/scripts/bench_flex.py 16 > $OUT/pmc.log 2>&1
cd $R
python - <<PY
import csv, glob, collections, json
f = glob.glob("$OUT/pmc/**/*counter_collection.csv", recursive=True)
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for r in csv.DictReader(open(f[0])):
    k = r["Kernel_Name"].split("(")[0][:60]
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == "SQ_WAVES": cnt[k] += 1
out = {}
for k, v in acc.items():
    if "dense" in k or "flex" in k or "linearize" in k:
        n = max(cnt[k], 1)
        out[k] = {"launches": cnt[k], **{c: x / n for c, x in v.items()}}
        if v.get("SQ_WAVE_CYCLES"):
            out[k]["valu_active_of_wave_cycles"] = v.get("SQ_ACTIVE_INST_VALU", 0) / v["SQ_WAVE_CYCLES"]
json.dump(out, open("$OUT/pmc_flex.json", "w"), indent=1); print(json.dumps(out, indent=1)[:3000])
PY
find $OUT -name '*.csv' -size +8M -delete
