#!/bin/bash
# k_geom_member / k_geom_design durations of library variants, alone on the GPU.  Usage: bash scripts/gpu_exp_member.sh <tag> head lib1.so ...
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
for v in "$@"; do
  n=$(basename $v .so)
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$n -o x -- python $R/scripts/exp_member.py $v > $OUT/$n.log 2>&1
  echo "== $v"; grep -E "k_geom_member|k_geom_design\(|k_geom_scan|k_geom_reduce" $OUT/$n/*kernel_stats.csv $OUT/$n/*/*kernel_stats.csv 2>/dev/null | cut -d, -f1-4 | sed 's/.*"k_geom/k_geom/'
done
find $OUT -name '*.csv' -size +8M -delete
