"""Per-phase cycle breakdown of the sweep kernel (needs raft_amd/csrc/libraftx_hip_timing.so: python raft_amd/csrc/build.py --timing)."""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.getcwd())
import bench
from raft_amd._abi import RaftxLib
lib = RaftxLib(os.environ.get("RAFTX_TIMING_LIB", os.path.join("raft_amd", "csrc", "libraftx_hip_timing.so")))
ctx = lib.context(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
sw, fx, geo = bench.make_sweep(ctx, n, 0, pinned=False)
sw.upload(ctx)                       # device-generated tables (the bench workload)
for _ in range(2):
    ctx.solve_dynamics_device(sw.nIter, sw.tol, sw.XiStart)
ms = ctx.last_kernel_ms()
out = (ctypes.c_ulonglong * 8)()
lib.lib.raftx_debug_phase_cycles.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
assert lib.lib.raftx_debug_phase_cycles(ctx._h, out) == 0
v = np.array(list(out), dtype=float)
names = ["setup+inertial", "passA reduce", "strip_phase", "passB", "solve+conv", "tail", "passA strips", "XiLast fetch"]
print("shape", os.environ.get("RAFTX_SHAPE", "default"), "kernel_ms %.3f" % ms, "cycles/workgroup %.0f" % (v.sum() / n))
for nm, x in zip(names, v):
    if x:
        print("  %-16s %6.1f %%  %10.0f cycles/workgroup" % (nm, 100 * x / v.sum(), x / n))
