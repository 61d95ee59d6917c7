"""Isolated crossings with the responses downloaded (scripts/gpu_r4_iso_trace.sh, gpu_r4_slab.sh): wall time per call, and
the responses' bits against the first call of the process."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hashlib
import numpy as np
import bench
from raft_amd import backend
ctx = backend.hip_library().context(0)
sw, fx, geo = bench.make_sweep(ctx, 10000, 0, pinned=True)
X = ctx.pinned_empty((10000, 1, 1, 6, sw.nw))
ts = []
for i in range(int(os.environ.get("ISO_CALLS", "8"))):
    X[...] = 0
    t0 = time.perf_counter()
    r = sw.run_crossing(ctx, Xi_out=X)
    ts.append(1e3 * (time.perf_counter() - t0))
    if i == 1:
        print("isolated xi call %.3f ms  (library wall %.3f)" % (ts[-1], r["timing_ms"][0]))
h = hashlib.sha256(np.ascontiguousarray(X).view(np.uint8)).hexdigest()[:16]
h2 = hashlib.sha256(np.ascontiguousarray(r["std"]).view(np.uint8)).hexdigest()[:16]
print("%s: median of calls 2.. %.3f ms, min %.3f; Xi sha %s std sha %s" % (os.environ.get("TAG", ""), float(np.median(ts[2:])), min(ts[2:]), h, h2))
