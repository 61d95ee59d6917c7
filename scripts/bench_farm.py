#!/usr/bin/env python
"""configs[3] at full size: the 4-unit VolturnUS-S farm (24-DOF coupled solve) x 50 sea states (nw of the committed
fixture), timed per launch.  One JSON line."""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from raft_amd import backend, dropin
from tests.util import load_model_fixture, case_from_fixture

fx, model = load_model_fixture("c4_farm.npz")
base = [case_from_fixture(c) for c in fx["cases"]]
rng = np.random.default_rng(1)
cases = []
for i in range(50):
    c = dict(base[i % 2])
    c["wave_height"], c["wave_period"], c["wave_heading"] = float(rng.uniform(1, 10)), float(rng.uniform(6, 16)), float(rng.uniform(0, 360))
    cases.append(c)
sweep = dropin.sweep_from_units(model, cases)
ctx = backend.default_context(0)
for _ in range(3):
    t0 = time.perf_counter()
    out = sweep.run_farm(ctx, 4, Cc=fx["coupling_C"][None])
    wall = time.perf_counter() - t0
nw = model.nw
print(json.dumps({"units": 4, "sea_states": 50, "nw": int(nw), "unit_fixed_points_kernel_ms": out["kernel_ms"][0],
                  "coupled_24x24_solves_kernel_ms": out["kernel_ms"][1], "wall_ms_incl_upload_and_download": 1e3 * wall,
                  "coupled_solves_per_s": 50 * nw / (out["kernel_ms"][1] * 1e-3), "converged_fraction": float(np.mean(out["flags"] & 1))}))
