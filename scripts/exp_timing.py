#!/usr/bin/env python
"""Kernel-time A/B on the bench workload WITHOUT parity checks (tuning experiments only: some variants compute wrong
numbers on purpose to bound what an optimisation could buy).  Usage: python scripts/exp_timing.py lib1.so lib2.so ...
('head' = the product library).  Prints per library: median kernel ms over the timed launches and the mean iteration
count (a variant whose iteration count differs from head's is not comparable)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench                                   # noqa: E402
from raft_amd._abi import RaftxLib             # noqa: E402
from raft_amd.backend import HIP_LIB_PATH      # noqa: E402

libs = sys.argv[1:] or ["head"]
reps = int(os.environ.get("EXP_REPS", "3"))
rows = {}
for rep in range(reps):
    for name in libs:
        path = HIP_LIB_PATH if name == "head" else os.path.join(ROOT, name)
        ctx = RaftxLib(path).context(0)
        sw, _, _ = bench.make_sweep(ctx, int(os.environ.get("EXP_DESIGNS", "10000")), 0, pinned=False)
        sw.upload(ctx)                          # device-generated tables + the sea state (the bench workload)
        ms = []
        for i in range(25):
            ctx.solve_dynamics_device(sw.nIter, float(os.environ.get("EXP_TOL", "0.01")), sw.XiStart)
            if i >= 5:
                ms.append(ctx.last_kernel_ms())
        res = ctx.fetch_results(want_Xi=False)
        rows.setdefault(name, []).append((float(np.median(ms)), float(res["niter"].mean())))
        ctx.close()
for name in libs:
    print(name, " ".join("%.3f ms (it %.3f)" % r for r in rows[name]))
