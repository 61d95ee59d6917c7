"""D2H of a few MB into FRESH NumPy arrays against a reused one (blocking fetch_results calls): where do the occasional
tens-of-ms calls come from?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from raft_amd import backend
ctx = backend.hip_library().context(0)
sw, fx, geo = bench.make_sweep(ctx, 600, 0, pinned=False)
sw.upload(ctx); sw.solve(ctx, upload=False)
shape = (600, 1, 1, 6, sw.nw)
def t(f, n=60):
    ts = []
    for _ in range(n):
        t0 = time.perf_counter(); f(); ts.append(1e3 * (time.perf_counter() - t0))
    ts = np.array(ts[5:]); return "median %.2f ms, max %.2f, calls over 5 ms: %d of %d" % (np.median(ts), ts.max(), int((ts > 5).sum()), len(ts))
reused = np.empty(shape, dtype=np.complex128)
print("fresh output each call :", t(lambda: ctx.fetch_results(want_Xi=True)))
print("reused output          :", t(lambda: ctx.fetch_results(want_Xi=True, Xi_out=reused)))
keep = []
print("fresh, kept alive      :", t(lambda: keep.append(ctx.fetch_results(want_Xi=True)["Xi"])))
pin = ctx.pinned_empty(shape)
print("page-locked output     :", t(lambda: ctx.fetch_results(want_Xi=True, Xi_out=pin)))
