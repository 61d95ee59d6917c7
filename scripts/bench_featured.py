#!/usr/bin/env python
"""The featured-sweep legs of bench.py alone (resident kernel timings of the lean featured specialisations against the
plain C3 sweep, a sample of each against the oracle).  Usage: python scripts/bench_featured.py [n_designs]"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench                                   # noqa: E402
from raft_amd import backend                   # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
ctx = backend.hip_library().context(0)
sw, _, _ = bench.make_sweep(ctx, n, 0, pinned=False)
sw.upload(ctx)
ks = []
for i in range(8):
    ctx.solve_dynamics_device(sw.nIter, sw.tol, sw.XiStart)
    if i >= 2:
        ks.append(ctx.last_kernel_ms())
niter = ctx.fetch_results(want_Xi=False)["niter"]
out = {"plain_sweep": {"kernel_ms": float(np.mean(ks)), "pairs": n, "kernel_flags": ctx.last_solve_kernel()[0],
                       "run_start_cache_slots": ctx.last_solve_kernel()[2],
                       "ns_per_pair_iteration": 1e6 * float(np.mean(ks)) / float(np.sum(niter))}}
out.update(bench.featured_legs(ctx, n, sw, float(np.mean(ks)), float(np.sum(niter))))
for k, v in out.items():
    print(k, json.dumps(v))
