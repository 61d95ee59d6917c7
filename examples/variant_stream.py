#!/usr/bin/env python
"""A STREAM of candidate batches of a VolturnUS-S parameter sweep on one MI355X, with no host work per candidate:

    python examples/variant_stream.py [n_designs_per_batch] [n_batches]

The base unit and the dependent-geometry edits of raft/parametersweep.py:56-87 go to the device ONCE as an edit program
(raft_amd.geometry.volturnus_program -> raftx_variant_program); after that every batch is its candidates' five parameters
(40 B per design over PCIe): the library writes the member descriptions in HBM, generates strip tables and statics, runs
every fixed point and sends back the motion statistics (raftx_sweep_prepare_variants / raftx_sweep_launch / raftx_sweep_wait,
two batches in flight).  An optimiser would draw the next batch's parameters from the statistics of the last one; here they
come from a random stream and the best candidate so far is tracked.

Runs on the committed fixtures (no reference tree needed)."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from raft_amd import backend, geometry as G                              # noqa: E402
from raft_amd.sweep import VariantSweep                                  # noqa: E402
from raft_amd import snapshot as standin                                 # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
    n_batches = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    fg = standin.load_fixture("geom_units.npz")
    c3 = standin.load_fixture("c3_variants.npz")
    base = json.loads(fg["c3_base_json"])
    u0 = [u for u in fg["units"] if u["name"] == "C3-variant-0"][0]
    M_rna = np.asarray(u0["M_struc"]) - np.asarray(u0["M_struc_bare"])             # rotor-nacelle assembly: not geometry
    C_rest = np.asarray(u0["C_struc"]) - np.asarray(u0["C_struc_bare"]) + np.diag([7e4, 7e4, 0, 0, 0, 1e8])   # + mooring
    rng = np.random.default_rng(2)
    draw = lambda: G.volturnus_params(rng.uniform(0.75, 1.25, size=(n, 5)))        # (ccD, ocD, T, ocR, pH) per candidate
    rep = lambda a: np.repeat(a[None], n, axis=0)
    sweep = VariantSweep(G.volturnus_program(base), draw(), rep(M_rna), np.zeros((n, 6, 6)), rep(C_rest), c3["w"], c3["k"],
                         float(c3["depth"]), np.asarray(c3["zeta"])[None], np.asarray(c3["beta"])[None], int(c3["nIter"]),
                         float(c3["XiStart"]))
    ctx = backend.default_context(0)
    for _ in range(4):                                                    # the first batches of a process also pay for its start:
        sweep.wait_crossing(ctx, sweep.submit_crossing(ctx, 0))           # code-object load, device buffers, the chip's clock ramp
        sweep.set_params(draw())
    best = (np.inf, None)
    params_in_flight = {}
    t0 = time.perf_counter()
    h = sweep.submit_crossing(ctx, 0)
    params_in_flight[0] = sweep.params
    for b in range(n_batches):
        h_next = None
        if b + 1 < n_batches:
            sweep.set_params(draw())                                      # the next candidates: drawn while this batch solves
            params_in_flight[(b + 1) % 2] = sweep.params
            h_next = sweep.submit_crossing(ctx, (b + 1) % 2)
        out = sweep.wait_crossing(ctx, h)
        pitch = np.where(out["flags"][:, 0] & 1, out["std"][:, 0, 4], np.inf)      # converged candidates only
        i = int(np.argmin(pitch))
        if pitch[i] < best[0]:
            best = (float(pitch[i]), params_in_flight[b % 2][i].copy())
        h = h_next
    dt = time.perf_counter() - t0
    nw = len(c3["w"])
    print("%d batches x %d new candidates x %d bins in %.1f ms: %.2f ms per batch, %.0f M design-case-frequency solves/s"
          % (n_batches, n, nw, 1e3 * dt, 1e3 * dt / n_batches, n_batches * n * nw / dt / 1e6))
    print("smallest pitch std %.3f deg for (ccD, ocD, T, ocR, pH) = %s" % (best[0], np.round(best[1], 2)))


if __name__ == "__main__":
    main()
