"""Per-phase cycles of the fused kernel per (pair, iteration) for launches of ONE residency round (every workgroup in step) and of
many rounds (phases mixed): which phases pay for being in step.  Timing build (python raft_amd/csrc/build.py --timing)."""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.getcwd())
import bench
from raft_amd._abi import RaftxLib
lib = RaftxLib(os.environ.get("RAFTX_TIMING_LIB", os.path.join("raft_amd", "csrc", "libraftx_hip_timing.so")))
lib.lib.raftx_debug_phase_cycles.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
names = ["setup+inertial", "passA reduce", "strip_phase", "passB", "solve+conv", "tail", "passA strips", "XiLast fetch"]
res = {}
for n in (1024, 2048, 10000, 30000):
    ctx = lib.context(0)
    sw, fx, geo = bench.make_sweep(ctx, n, 0, pinned=False)
    sw.upload(ctx)
    reps = 4
    for _ in range(reps):
        ctx.solve_dynamics_device(sw.nIter, sw.tol, sw.XiStart)
    ms = ctx.last_kernel_ms()
    r = ctx.fetch_results(want_Xi=False)
    out = (ctypes.c_ulonglong * 8)()
    assert lib.lib.raftx_debug_phase_cycles(ctx._h, out) == 0
    pit = float(np.sum(r["niter"]))
    v = np.array(list(out), dtype=float) / pit                 # (the counters are zeroed at every launch: the last one's)
    res[n] = v
    print("n = %5d: kernel %.3f ms, %.1f ns per (pair, iteration); cycles per (pair, iteration): %s" %
          (n, ms, 1e6 * ms / pit, "  ".join("%s %.0f" % (nm, x) for nm, x in zip(names, v) if x)))
    ctx.close()
print("ratio 1024 / 30000:", {nm: round(float(a / b), 3) for nm, a, b in zip(names, res[1024], res[30000]) if b})
