#!/usr/bin/env python
"""configs[3] at full size: the 4-unit VolturnUS-S farm (24-DOF coupled solve) x 200 bins x 50 sea states, timed per launch and
checked against the live reference's responses of all 50 sea states.  One JSON line."""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from raft_amd import backend, dropin
from raft_amd.snapshot import load_model_fixture, case_from_fixture, ref_headings
from raft_amd.metrics import group_rel_err

fx, model = load_model_fixture("c4_farm.npz")            # nw = 200, the 50 seeded sea states of default_rng(1)
cases = [case_from_fixture(c) for c in fx["cases"]]
assert len(cases) == 50 and model.nw == 200
sweep = dropin.sweep_from_units(model, cases)
ctx = backend.default_context(0)
for _ in range(3):
    t0 = time.perf_counter()
    out = sweep.run_farm(ctx, 4, Cc=fx["coupling_C"][None])
    wall = time.perf_counter() - t0
nw = model.nw
err = max(group_rel_err(out["Xi"][0, i, :1], ref_headings(c)[0]) for i, c in enumerate(fx["cases"]))
print(json.dumps({"max_group_rel_err_vs_live_reference_all_50_sea_states": err,
                  "units": 4, "sea_states": 50, "nw": int(nw), "unit_fixed_points_kernel_ms": out["kernel_ms"][0],
                  "coupled_24x24_solves_kernel_ms": out["kernel_ms"][1], "wall_ms_incl_upload_and_download": 1e3 * wall,
                  "coupled_solves_per_s": 50 * nw / (out["kernel_ms"][1] * 1e-3), "converged_fraction": float(np.mean(out["flags"] & 1))}))
