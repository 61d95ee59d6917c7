"""Summarise a gpu_round.sh output directory into profiles/<name>/ (kernel stats + per-launch PMC means)."""
import csv, collections, json, os, shutil, sys
src, dst = sys.argv[1], sys.argv[2]
os.makedirs(dst, exist_ok=True)
for f in ("trace/bench_kernel_stats.csv", "bench.json"):
    if os.path.exists(os.path.join(src, f)):
        shutil.copy(os.path.join(src, f), os.path.join(dst, os.path.basename(f).replace("bench_", "")))
out = {}
for d in sorted(os.listdir(src)):
    p = os.path.join(src, d, "bench_counter_collection.csv")
    if not os.path.exists(p):
        continue
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(p)):
        if "k_solve_dynamics" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
            out["_kernel"] = r["Kernel_Name"]
            out["_vgpr"], out["_agpr"], out["_scratch"] = r["VGPR_Count"], r["Accum_VGPR_Count"], r["Scratch_Size"]
            out["_wg"], out["_grid"] = r["Workgroup_Size"], r["Grid_Size"]
    for k, v in acc.items():
        out[k] = {"launches": len(v), "mean_per_launch": sum(v) / len(v)}
json.dump(out, open(os.path.join(dst, "pmc_summary.json"), "w"), indent=1)
for k, v in out.items():
    print(k, v if not isinstance(v, dict) else "%.4g" % v["mean_per_launch"])
