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
    rows = [r for r in csv.DictReader(open(p)) if "k_solve_dynamics" in r["Kernel_Name"] or "raftx_kp_f" in r["Kernel_Name"]]
    # whole-batch launches only: the closing parity crossing downloads its responses and is cut into slabs of one residency round
    gmax = max((int(r["Grid_Size"]) for r in rows), default=0)
    for r in rows:
        if int(r["Grid_Size"]) == gmax:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
            out["_kernel"] = r["Kernel_Name"]
            out["_vgpr"], out["_agpr"], out["_scratch"] = r["VGPR_Count"], r["Accum_VGPR_Count"], r["Scratch_Size"]
            out["_wg"], out["_grid"] = r["Workgroup_Size"], r["Grid_Size"]
    for k, v in acc.items():
        out[k] = {"launches": len(v), "mean_per_launch": sum(v) / len(v)}
out["_note"] = ("_vgpr / _agpr are rocprofv3's VGPR_Count / Accum_VGPR_Count columns: on gfx950 (unified 512-entry file) the tool reports "
                "the ARCH-VGPR allocation in its own granule (128 for every kernel compiled to two waves per SIMD), not the compiler's "
                "count -- k_solve_dynamics<2,0,128,2> is 254 VGPRs + 0 AGPRs, 0 B scratch by -Rpass-analysis=kernel-resource-usage (DESIGN.md 3.1)")
json.dump(out, open(os.path.join(dst, "pmc_summary.json"), "w"), indent=1)
if "FETCH_SIZE" in out and "WRITE_SIZE" in out:      # rocprofv3 reports KB; FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950
    f, w_ = out["FETCH_SIZE"]["mean_per_launch"] * 1e3, out["WRITE_SIZE"]["mean_per_launch"] * 1e3
    grid, wg = int(out.get("_grid", 0) or 0), int(out.get("_wg", 1) or 1)
    t = {"shape": "2x128", "designs_per_launch": grid // wg if wg else 0, "kernel": "k_solve_dynamics", "FETCH_SIZE_bytes_raw": f,
         "WRITE_SIZE_bytes_raw": w_, "hbm_bytes_per_launch": 2.0 * f + w_, "source": dst + "/pmc_summary.json (scripts/gpu_round.sh: separate "
         "--pmc FETCH_SIZE / WRITE_SIZE passes, whole-batch launches: bench.py --chunks 1)",
         "note": "FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 counts 128-B requests as 64 B); WRITE_SIZE uncalibrated; "
                 "Infinity-Cache hits are counted as traffic"}
    import subprocess, time
    try:                                                 # which kernel source the counters belong to, and when they were summarised
        t["kernel_source_head"] = subprocess.run(["git", "log", "-1", "--format=%h %s", "--", "raft_amd/csrc"], capture_output=True, text=True).stdout.strip()
    except OSError:
        pass
    t["measured_at"] = time.strftime("%Y-%m-%d %H:%M")
    json.dump(t, open(os.path.join(dst, "traffic.json"), "w"), indent=1)
    json.dump(t, open(os.path.join("profiles", "traffic_latest.json"), "w"), indent=1)
for k, v in out.items():
    print(k, v if not isinstance(v, dict) else "%.4g" % v["mean_per_launch"])
