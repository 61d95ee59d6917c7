#!/usr/bin/env python
"""A batch of floating wind turbines with FLEXIBLE members on one MI355X: the reference's VolturnUS-S-flexible deck (beam
pontoons and a beam tower: 150 reduced degrees of freedom) in several stiffness variants x several sea states, every
(unit, sea state) fixed point on the device in one call:

    python examples/flex_sweep.py [n_units]

1. the live model comes from the committed fixture (tests/golden/flex_volturnus.npz: the reference's own T reduction,
   structural, hydrostatic and elastic matrices -- the finite-element assembly stays upstream);
2. host: strip tables per wet structural node (raft_amd.strips.pack_fowt_nodes), the inertial excitation reduced once;
3. device (raftx_flex_solve): node motions, the drag linearisation of every node, the projections with T (MFMA tiles), the
   150 x 150 impedance solves of every unit, sea state and frequency bin in one launch per iteration, the convergence test
   and relaxation per (unit, sea state) -- a pair that has converged is frozen, as if it had been solved alone.

What the reference does one load case at a time in about a second (Model.solveDynamics)."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from raft_amd import backend, dropin                                     # noqa: E402
from raft_amd.flex import FlexSweep, FlexUnit                            # noqa: E402
from raft_amd.snapshot import load_model_fixture, case_from_fixture      # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    fx, model = load_model_fixture("flex_volturnus.npz")
    base = case_from_fixture(fx["cases"][0])
    cases = [base, dict(base, wave_height=4.0, wave_period=9.0, wave_heading=-20.0), dict(base, wave_height=1.0, wave_period=6.0)]
    one = dropin.flex_sweep_from_models([model], cases)                  # sea states, frequency grid, the unit as it is
    u0 = one.units[0]
    stiff = np.linspace(0.7, 1.3, n)                                     # variants: the elastic + hydrostatic stiffness scaled
    units = [FlexUnit(u0.tables, u0.Tn, u0.M, u0.B, u0.C * s) for s in stiff]
    sweep = FlexSweep(units, one.w, one.k, one.depth, one.zeta, one.beta, one.nIter, one.XiStart, one.tol)
    ctx = backend.default_context(0)
    sweep.run(ctx)                                                       # first call of the process: library start-up
    t0 = time.perf_counter()
    out = sweep.run(ctx)
    dt = time.perf_counter() - t0
    Xi = out["Xi"]                                                       # [unit, sea state, heading, DOF, bin]
    dw = float(one.w[1] - one.w[0])
    surge = np.sqrt(np.sum(np.abs(Xi[:, :, 0, 0, :]) ** 2, axis=-1) * 0.5 / dw * dw)       # std of surge, as getRMS
    print("%d units x %d sea states (150 DOFs, %d bins): %.1f ms = %.2f ms per (unit, sea state); fixed point on the device %.1f ms"
          % (n, len(cases), len(one.w), 1e3 * dt, 1e3 * dt / (n * len(cases)), out["kernel_ms"][1]))
    print("iterations per sea state (first unit): %s of at most %d; converged within the tolerance: %.0f %% of the pairs"
          % (out["niter"][0].tolist(), one.nIter + 1, 100.0 * float((out["flags"] & 1).mean())))
    for i in (0, n // 2, n - 1):
        print("  stiffness x %.2f: surge std per sea state %s m" % (stiff[i], np.round(surge[i], 4).tolist()))


if __name__ == "__main__":
    main()
