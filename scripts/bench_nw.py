#!/usr/bin/env python
"""Resident fused-kernel time of the C3 designs at other frequency counts (the launch shapes beyond 256 bins):
python scripts/bench_nw.py [n_designs] [nw ...]   -> dcf/s per shape, the kernel specialisation that ran."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench                                   # noqa: E402
from raft_amd import backend, waves            # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
nws = [int(a) for a in sys.argv[2:]] or [200, 256, 400, 512, 1000, 2000]
ctx = backend.hip_library().context(0)
for nw in nws:
    sw, fx, _ = bench.make_sweep(ctx, n, 0, pinned=False)
    w = np.linspace(float(fx["w"][0]), float(fx["w"][-1]), nw)
    dw = float(w[1] - w[0])
    sw.w, sw.k = w, np.array([waves.wave_number(x, sw.depth) for x in w])
    sw.zeta = np.sqrt(2.0 * waves.jonswap(w, 6.0, 12.0) * dw)[None, None, :]
    sw.upload(ctx)
    ks = []
    for i in range(6):
        ctx.solve_dynamics_device(sw.nIter, sw.tol, sw.XiStart)
        if i >= 2:
            ks.append(ctx.last_kernel_ms())
    r = ctx.fetch_results(want_Xi=False)
    fl, wps, slots = ctx.last_solve_kernel()
    print(json.dumps({"nw": nw, "designs": n, "kernel_ms": float(np.mean(ks)), "dcf_per_s": n * nw / (np.mean(ks) * 1e-3),
                      "mean_iterations": float(np.mean(r["niter"])), "flags": fl, "waves_per_simd": wps, "cache_slots": slots,
                      "nan": int(np.count_nonzero(r["flags"] & 2))}))
