"""gpurun_out/<tag>/<shape>_{FETCH,WRITE}_SIZE/b_counter_collection.csv -> profiles/traffic_latest.json
usage: python scripts/traffic_summary.py gpurun_out/t5 2x128 10000 profiles/r01_v4d"""
import csv, json, os, sys
src, shape, designs, dst = sys.argv[1], sys.argv[2], int(sys.argv[3]), sys.argv[4]
vals = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    p = os.path.join(src, "%s_%s" % (shape, c), "b_counter_collection.csv")
    v = [float(r["Counter_Value"]) for r in csv.DictReader(open(p)) if "k_solve_dynamics" in r["Kernel_Name"] or "raftx_kp_f" in r["Kernel_Name"]]
    vals[c] = sum(v) / len(v) * 1e3          # rocprofv3 reports KB
out = {"shape": shape, "designs_per_gpu": designs, "kernel": "k_solve_dynamics",
       "FETCH_SIZE_bytes_raw": vals["FETCH_SIZE"], "WRITE_SIZE_bytes_raw": vals["WRITE_SIZE"],
       "hbm_bytes_per_launch": 2.0 * vals["FETCH_SIZE"] + vals["WRITE_SIZE"],
       "note": "separate --pmc passes; FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 counts 128-B requests as 64 B); "
               "WRITE_SIZE uncalibrated; Infinity-Cache hits are counted as traffic"}
os.makedirs(dst, exist_ok=True)
for path in (os.path.join(dst, "traffic.json"), os.path.join("profiles", "traffic_latest.json")):
    json.dump(out, open(path, "w"), indent=1)
print(json.dumps(out))
