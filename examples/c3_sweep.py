#!/usr/bin/env python
"""A VolturnUS-S geometry sweep end to end on one MI355X, from member descriptions to response statistics:

    python examples/c3_sweep.py [n_designs]

1. host: the baseline design's members are parsed once (raft_amd.geometry.describe_unit) and the five sweep parameters of
   raft/parametersweep.py are applied to the descriptor arrays with NumPy broadcasting (raft_amd/geometry.py volturnus_sweep);
2. device: strip tables, Morison added mass, hydrostatics, member inertia -- with the ballast density trimmed for heave
   equilibrium (Model.adjustBallastDensity) -- are generated for all designs (raftx_build_designs);
3. device: the drag-linearised frequency-domain responses of every design in three sea states (one launch);
4. device: motion statistics; only ~150 bytes per (design, sea state) come back.

Runs on the committed fixtures (no reference tree needed)."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from raft_amd import backend, geometry as G, waves                      # noqa: E402
from raft_amd.sweep import GeometrySweep                                 # noqa: E402
from raft_amd import snapshot as standin                                                # noqa: E402
from raft_amd.geometry import volturnus_sweep                                   # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
    fg = standin.load_fixture("geom_units.npz")
    c3 = standin.load_fixture("c3_variants.npz")
    base = json.loads(fg["c3_base_json"])
    u0 = [u for u in fg["units"] if u["name"] == "C3-variant-0"][0]
    M_rna = np.asarray(u0["M_struc"]) - np.asarray(u0["M_struc_bare"])             # rotor-nacelle assembly: not geometry
    C_rest = np.asarray(u0["C_struc"]) - np.asarray(u0["C_struc_bare"]) + np.diag([7e4, 7e4, 0, 0, 0, 1e8])   # + mooring
    scales = np.random.default_rng(1).uniform(0.8, 1.2, size=(n, 5))
    t0 = time.perf_counter()
    tables = volturnus_sweep(base, scales).tables()
    t_desc = time.perf_counter() - t0
    w, k, depth = np.asarray(c3["w"]), np.asarray(c3["k"]), float(c3["depth"])
    dw = w[1] - w[0]
    seas = [(6.0, 12.0), (2.0, 8.0), (10.0, 14.0)]
    zeta = np.array([[np.sqrt(2 * waves.jonswap(w, Hs, Tp) * dw)] for Hs, Tp in seas])          # [nCase, nHead=1, nw]
    beta = np.zeros((len(seas), 1))
    sweep = GeometrySweep(tables, np.repeat(M_rna[None], n, 0), np.zeros((n, 6, 6)), np.repeat(C_rest[None], n, 0),
                          w, k, depth, zeta, beta, nIter=int(c3["nIter"]), XiStart=float(c3["XiStart"]),
                          add_mask=G.ADD_MORISON | G.ADD_HYDROSTATIC | G.ADD_INERTIA | G.TRIM_BALLAST)
    ctx = backend.default_context(0)
    t0 = time.perf_counter()
    out = sweep.run_stats(ctx)
    t_all = time.perf_counter() - t0
    S = ctx.fetch_statics()
    ok = (out["flags"] & 1).astype(bool)
    print("%d designs x %d sea states x %d bins: descriptors %.1f ms (host), generate + solve + statistics %.1f ms"
          % (n, len(seas), len(w), 1e3 * t_desc, 1e3 * t_all))
    print("converged %.1f %% of (design, sea state) pairs; mean iterations %.2f" % (100 * ok.mean(), out["niter"].mean()))
    print("ballast density trim: %.0f .. %.0f kg/m^3; displaced volume %.0f .. %.0f m^3"
          % (S["props"][:, G.SP_DRHO].min(), S["props"][:, G.SP_DRHO].max(), S["props"][:, G.SP_V].min(), S["props"][:, G.SP_V].max()))
    i = int(np.argmin(out["std"][:, 0, 4]))
    print("smallest pitch std in the (6 m, 12 s) sea: %.3f deg for scales %s" % (out["std"][i, 0, 4], np.round(scales[i], 3)))


if __name__ == "__main__":
    main()
