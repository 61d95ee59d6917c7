#!/usr/bin/env python
"""Times the device geometry generator (raftx_build_designs) on a C3-style sweep: the member descriptors of the
committed reference-built variants (tests/golden/geom_units.npz, C3-variant-*) tiled to N designs.  Prints one
JSON line: device time of the five kernels (HIP events), wall time of the whole call (descriptor H2D included),
and, for comparison, the wall time of raftx_upload_designs of the SAME strip tables (256 B/strip H2D + host run
detection) -- the path a host-side packer has to take.  Writes profiles/<tag>_geom_bench.json when --tag is given."""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--designs", type=int, default=10000)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--tag", default="")
    args = ap.parse_args()
    from raft_amd import backend, geometry as G
    from raft_amd import snapshot as standin
    fx = standin.load_fixture("geom_units.npz")
    units = [u for u in fx["units"] if u["name"].startswith("C3-variant")]
    t0 = time.perf_counter()
    tabs = [G.describe_unit(json.loads(u["design_json"])) for u in units]
    t_parse = (time.perf_counter() - t0) / len(units)
    nD = args.designs
    tiled = [tabs[i % len(tabs)] for i in range(nD)]
    D = G.concat_units(tiled)
    mo, mem, so, st = D
    nw = len(units[0]["w"])
    Z = np.zeros((nD, 6, 6))
    ctx = backend.hip_library().context(0)
    walls, devs = [], []
    for _ in range(args.reps):
        t0 = time.perf_counter()
        off = ctx.build_designs(mo, mem, so, st, Z, Z, Z, nw, rho=1025.0, g=9.81, cap_off=D.cap_off, caps=D.caps,
                                add_mask=G.ADD_MORISON | G.ADD_HYDROSTATIC | G.ADD_INERTIA)
        walls.append(time.perf_counter() - t0)
        devs.append(ctx.last_kernel_ms())
    strips, _ = ctx.fetch_strips(off[-1])
    up = []
    for _ in range(args.reps):
        t0 = time.perf_counter()
        ctx.upload_designs_raw(off, strips, Z, Z, Z, nw)
        up.append(time.perf_counter() - t0)
    ctx.close()
    out = {"designs": nD, "members": int(mo[-1]), "stations": int(so[-1]), "strips": int(off[-1]),
           "descriptor_bytes": int(mem.nbytes + st.nbytes + mo.nbytes + so.nbytes + D.caps.nbytes + D.cap_off.nbytes), "strip_table_bytes": int(strips.nbytes),
           "device_ms": float(np.median(devs)), "build_designs_wall_ms": 1e3 * float(np.median(walls)),
           "upload_designs_wall_ms": 1e3 * float(np.median(up)), "host_parse_ms_per_design": 1e3 * t_parse,
           "designs_per_s_device": nD / (1e-3 * float(np.median(devs)))}
    print(json.dumps(out))
    if args.tag:
        with open(os.path.join(ROOT, "profiles", args.tag + "_geom_bench.json"), "w") as f:
            json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
