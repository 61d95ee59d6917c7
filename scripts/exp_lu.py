#!/usr/bin/env python
"""Timing only (no checks): the farm sweep's two kernels for a tuning build of the library (RAFTX_HIP_LIB=...)."""
import json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from raft_amd import backend, dropin
from raft_amd.snapshot import load_model_fixture, case_from_fixture
from raft_amd.sweep import Sweep
fx, model = load_model_fixture("c4_farm.npz")
cases = [case_from_fixture(c) for c in fx["cases"]]
sweep = dropin.sweep_from_units(model, cases)
ctx = backend.default_context(0)
nF = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
rep = lambda a: None if a is None else np.concatenate([a] * nF, axis=0)
off = np.concatenate([[0]] + [sweep.off[1:] + i * sweep.off[-1] for i in range(nF)])
big = Sweep(off, rep(sweep.strips), rep(sweep.M0), rep(sweep.B0), rep(sweep.C0), sweep.w, sweep.k, sweep.depth, sweep.zeta, sweep.beta,
            sweep.nIter, sweep.XiStart, tol=sweep.tol, MBw=rep(sweep.MBw))
ks = []
for _ in range(4):
    ob = big.run_farm(ctx, 4, Cc=np.repeat(fx["coupling_C"][None], nF, axis=0))
    ks.append([float(x) for x in ob["kernel_ms"]])
print(json.dumps({"lib": os.environ.get("RAFTX_HIP_LIB", "default"), "farms": nF, "kernel_ms": ks[1:],
                  "finite": bool(np.isfinite(ob["Xi"]).all())}))
