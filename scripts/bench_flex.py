"""Batched flexible sweep (raft_amd/flex.py) on the GPU: N copies of the reference's flexible deck (150 reduced DOFs, 40 bins)
x 3 sea states in one batch, against the drop-in's one-case-at-a-time Model.solveDynamics.  One JSON line."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from raft_amd import backend, dropin
from raft_amd.snapshot import load_model_fixture, case_from_fixture
from raft_amd.metrics import rel_err

n_unit = int(sys.argv[1]) if len(sys.argv) > 1 else 32
fx, model = load_model_fixture("flex_volturnus.npz")
ctx = backend.default_context(0)
eng = dropin.Engine(ctx)
base = case_from_fixture(fx["cases"][0])
cases = [base, dict(base, wave_height=4.0, wave_period=9.0, wave_heading=-20.0), dict(base, wave_height=1.0, wave_period=6.0)]
single = []
for rep in range(2):
    t0 = time.perf_counter()
    single = [eng.solveDynamics(model, dict(c)).copy() for c in cases]
    t_single = (time.perf_counter() - t0) / len(cases)
sw = dropin.flex_sweep_from_models([model] * n_unit, cases)
ts = []
for rep in range(6):                                          # median of five (host-side jitter of a few ms per call on the GPU box)
    t0 = time.perf_counter()
    out = sw.run(ctx)
    ts.append(time.perf_counter() - t0)
t_batch = float(np.median(ts[1:]))
err = max(rel_err(out["Xi"][d, ic, 0], single[ic][0]) for d in range(n_unit) for ic in range(3))
pairs = n_unit * 3
print(json.dumps({"units": n_unit, "cases": 3, "dofs": 150, "nw": int(model.nw), "dropin_ms_per_unit_case": 1e3 * t_single,
                  "batch_ms": 1e3 * t_batch, "batch_ms_per_unit_case": 1e3 * t_batch / pairs, "speedup": t_single * pairs / t_batch,
                  "kernel_ms_strips_dense": out["kernel_ms"], "max_rel_err_vs_dropin": err, "iterations": out["niter"][0].tolist()}))
