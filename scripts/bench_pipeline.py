#!/usr/bin/env python
"""PCIe-inclusive throughput of the host-buffer boundary with and without copy/compute overlap (run_pipelined):
descriptors in -> full responses (19 KB per design-case) or statistics out.  One JSON line."""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from raft_amd import backend, geometry as G
from raft_amd.sweep import GeometrySweep, Pipeline
from raft_amd import snapshot as standin
from raft_amd.geometry import volturnus_sweep

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
fg = standin.load_fixture("geom_units.npz")
c3 = standin.load_fixture("c3_variants.npz")
u0 = [u for u in fg["units"] if u["name"] == "C3-variant-0"][0]
M_rna = np.asarray(u0["M_struc"]) - np.asarray(u0["M_struc_bare"])
C_rest = np.asarray(u0["C_struc"]) - np.asarray(u0["C_struc_bare"]) + np.diag([7e4, 7e4, 0, 0, 0, 1e8])
scales = np.random.default_rng(0).uniform(0.75, 1.25, size=(n, 5))
tables = volturnus_sweep(json.loads(fg["c3_base_json"]), scales).tables()
sweep = GeometrySweep(tables, np.repeat(M_rna[None], n, 0), np.zeros((n, 6, 6)), np.repeat(C_rest[None], n, 0), c3["w"], c3["k"],
                      float(c3["depth"]), c3["zeta"], c3["beta"], int(c3["nIter"]), float(c3["XiStart"]))
lib = backend.hip_library()
nw = len(c3["w"])
out = {"designs": n, "nw": nw}
ctx = lib.context(0)
ref = sweep.run(ctx)
for label, fn in (("serial_full", lambda: sweep.run(ctx)), ("serial_stats", lambda: sweep.run_stats(ctx))):
    fn()
    t0 = time.perf_counter(); fn(); out[label + "_Mdcf_s"] = n * nw / (time.perf_counter() - t0) / 1e6
Xp = ctx.pinned_empty(ref["Xi"].shape)
for _ in range(2):
    t0 = time.perf_counter()
    sweep.solve(ctx); ctx.fetch_results(Xi_out=Xp)
    out["serial_full_pinned_Mdcf_s"] = n * nw / (time.perf_counter() - t0) / 1e6
assert np.array_equal(Xp.view(np.uint64), ref["Xi"].view(np.uint64))
ctx.free_pinned(Xp)
ctx.close()
for workers, chunks in ((2, 4), (2, 8), (3, 6), (4, 8)):
    pipe = Pipeline(lib, workers)
    for fetch in ("Xi", "stats"):
        pipe.run(sweep, chunks, fetch=fetch)
        pipe.run(sweep, chunks, fetch=fetch)
        t0 = time.perf_counter()
        got = pipe.run(sweep, chunks, fetch=fetch)
        out["pipelined_%s_w%d_c%d_Mdcf_s" % (fetch, workers, chunks)] = n * nw / (time.perf_counter() - t0) / 1e6
        if fetch == "Xi":
            assert np.array_equal(got["Xi"].view(np.uint64), ref["Xi"].view(np.uint64))
    pipe.run(sweep, chunks, pinned=True)
    t0 = time.perf_counter()
    got = pipe.run(sweep, chunks, pinned=True)
    out["pipelined_Xi_pinned_w%d_c%d_Mdcf_s" % (workers, chunks)] = n * nw / (time.perf_counter() - t0) / 1e6
    assert np.array_equal(got["Xi"].view(np.uint64), ref["Xi"].view(np.uint64))
    pipe.close()
print(json.dumps(out))
