"""Five isolated crossings with the responses downloaded (for scripts/gpu_r4_iso_trace.sh)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from raft_amd import backend
ctx = backend.hip_library().context(0)
sw, fx, geo = bench.make_sweep(ctx, 10000, 0, pinned=True)
X = ctx.pinned_empty((10000, 1, 1, 6, sw.nw))
for i in range(6):
    t0 = time.perf_counter()
    r = sw.run_crossing(ctx, Xi_out=X)
    print("isolated xi call %.3f ms  (library wall %.3f)" % (1e3 * (time.perf_counter() - t0), r["timing_ms"][0]))
