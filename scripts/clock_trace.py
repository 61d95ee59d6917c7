"""Shader clock the workgroups of ONE fused launch lived at, by when they started (timing build: raftx_debug_clock_trace)."""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.getcwd())
import bench
from raft_amd._abi import RaftxLib
lib = RaftxLib(os.environ.get("RAFTX_TIMING_LIB", os.path.join("raft_amd", "csrc", "libraftx_hip_timing.so")))
lib.lib.raftx_debug_clock_trace.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
NB = 64
for n in [int(a) for a in sys.argv[1:]] or [10000, 30000]:
    ctx = lib.context(0)
    sw, fx, geo = bench.make_sweep(ctx, n, 0, pinned=False)
    sw.upload(ctx)
    for _ in range(3):
        ctx.solve_dynamics_device(sw.nIter, sw.tol, sw.XiStart)
    ms = ctx.last_kernel_ms()
    out = (ctypes.c_ulonglong * (2 * NB))()
    assert lib.lib.raftx_debug_clock_trace(ctx._h, out, NB) == 0
    v = np.array(list(out), dtype=float).reshape(NB, 2)
    print("launch of %d pairs, %.3f ms: shader clock [GHz] of the workgroups that started in each 125 us of the launch" % (n, ms))
    row = []
    for b in range(NB):
        if v[b, 1] > 0:
            row.append("%.2f:%.2f" % (0.125 * b, v[b, 0] / v[b, 1] * 0.1))
    print("   " + "  ".join(row))
    ctx.close()
