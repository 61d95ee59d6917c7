#!/usr/bin/env python
"""configs[3] at full size: the 4-unit VolturnUS-S farm (24-DOF coupled solve) x 200 bins x 50 sea states, timed per launch and
checked against the live reference's responses of all 50 sea states.  One JSON line.
`--sweep N`: a farm SWEEP -- N farms (replicas of the C4 layout: the coupled-solve kernel does not care that they are equal)
x 50 sea states: 4 N x 50 unit fixed points with resident Z / F_wave, then 50 N x 200 coupled 24 x 24 solves in one launch;
every farm's responses equal farm 0's."""
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
FLOP_PER_SOLVE = 8.0 * (24 ** 3 / 3.0 + 24 ** 2)      # pivoted complex LU (n^3/3 multiply-adds) + the two triangular sweeps, 8 flops per complex multiply-add
extra = {}
if "--sweep" in sys.argv:
    from raft_amd.sweep import Sweep
    nF = int(sys.argv[sys.argv.index("--sweep") + 1])
    rep = lambda a: None if a is None else np.concatenate([a] * nF, axis=0)
    off = np.concatenate([[0]] + [sweep.off[1:] + i * sweep.off[-1] for i in range(nF)])
    big = Sweep(off, rep(sweep.strips), rep(sweep.M0), rep(sweep.B0), rep(sweep.C0), sweep.w, sweep.k, sweep.depth, sweep.zeta, sweep.beta,
                sweep.nIter, sweep.XiStart, tol=sweep.tol, MBw=rep(sweep.MBw))
    for _ in range(2):
        ob = big.run_farm(ctx, 4, Cc=np.repeat(fx["coupling_C"][None], nF, axis=0))
    assert np.array_equal(ob["Xi"][0].view(np.uint64), ob["Xi"][nF - 1].view(np.uint64))
    assert max(group_rel_err(ob["Xi"][nF // 2, i, :1], ref_headings(c)[0]) for i, c in enumerate(fx["cases"])) < 1e-9
    n_solve = nF * 50 * nw
    extra = {"farm_sweep": {"farms": nF, "unit_pairs": 4 * nF * 50, "unit_fixed_points_kernel_ms": ob["kernel_ms"][0],
                            "coupled_solves": n_solve, "coupled_solves_kernel_ms": ob["kernel_ms"][1],
                            "coupled_solves_per_s": n_solve / (ob["kernel_ms"][1] * 1e-3),
                            "coupled_solve_tflops": n_solve * FLOP_PER_SOLVE / (ob["kernel_ms"][1] * 1e-3) / 1e12}}
print(json.dumps({"max_group_rel_err_vs_live_reference_all_50_sea_states": err, **extra,
                  "units": 4, "sea_states": 50, "nw": int(nw), "unit_fixed_points_kernel_ms": out["kernel_ms"][0],
                  "coupled_24x24_solves_kernel_ms": out["kernel_ms"][1], "wall_ms_incl_upload_and_download": 1e3 * wall,
                  "coupled_solves_per_s": 50 * nw / (out["kernel_ms"][1] * 1e-3), "converged_fraction": float(np.mean(out["flags"] & 1))}))
