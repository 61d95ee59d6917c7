cd /tmp; export TMPDIR=/tmp
for v in "" v_gdexp1 v_gdexp3 v_gdexp4; do
  lib=$GRAFT_REPO_ROOT/raft_amd/csrc/libraftx_hip.so; [ -n "$v" ] && lib=$GRAFT_REPO_ROOT/raft_amd/csrc/libraftx_hip_$v.so
  rm -rf /tmp/ge
  RAFTX_HIP_LIB=$lib rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ge -o g -- python $GRAFT_REPO_ROOT/scripts/bench_geom.py --reps 4 > /dev/null 2>&1
  python - "$v" <<'PY'
import csv, sys
rows = {r["Name"][:22]: float(r["AverageNs"]) / 1e3 for r in csv.DictReader(open("/tmp/ge/g_kernel_stats.csv"))}
print("%-10s" % (sys.argv[1] or "default"), " ".join("%s=%.1f" % (k, v) for k, v in rows.items() if k.startswith("k_geom_design")))
PY
done
