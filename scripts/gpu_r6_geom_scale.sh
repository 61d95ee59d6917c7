cd /tmp; export TMPDIR=/tmp
for n in 1250 2500 5000 10000 20000; do
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/gs_$n -o g -- python $GRAFT_REPO_ROOT/scripts/bench_geom.py --designs $n --reps 4 > /dev/null 2>&1
  python - $n <<'PY'
import csv, sys
n = sys.argv[1]
rows = {r["Name"][:22]: float(r["AverageNs"]) / 1e3 for r in csv.DictReader(open("/tmp/gs_%s/g_kernel_stats.csv" % n))}
print(n, " ".join("%s=%.1f" % (k, v) for k, v in rows.items() if k.startswith(("k_geom", "void k_geom"))))
PY
done
