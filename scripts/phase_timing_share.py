"""Where the three-sea-state sweep gains over the one-sea-state sweep: per-phase cycles of the fused kernel per (pair, iteration),
timing build (python raft_amd/csrc/build.py --timing).  Cases: A one sea state; B three sea states (the design's table shared by
three pairs that run side by side on one XCD); C three IDENTICAL sea states (same work per pair as A, tables shared as in B)."""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.getcwd())
import bench
from raft_amd import waves
from raft_amd._abi import RaftxLib
lib = RaftxLib(os.environ.get("RAFTX_TIMING_LIB", os.path.join("raft_amd", "csrc", "libraftx_hip_timing.so")))
lib.lib.raftx_debug_phase_cycles.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
names = ["setup+inertial", "passA reduce", "strip_phase", "passB", "solve+conv", "tail", "passA strips", "XiLast fetch"]


def run(tag, **kw):
    ctx = lib.context(0)
    sw, fx, geo = bench.make_sweep(ctx, n, 0, pinned=False, **kw)
    sw.upload(ctx)
    for _ in range(3):
        ctx.solve_dynamics_device(sw.nIter, sw.tol, sw.XiStart)
    ms = ctx.last_kernel_ms()
    r = ctx.fetch_results(want_Xi=False)
    out = (ctypes.c_ulonglong * 8)()
    assert lib.lib.raftx_debug_phase_cycles(ctx._h, out) == 0
    v = np.array(list(out), dtype=float)                  # (the counters are zeroed at every launch: the last one's)
    pit = float(np.sum(r["niter"]))
    print("%s: pairs %d, mean iterations %.3f, kernel %.3f ms, %.1f ns per (pair, iteration), cycles per (pair, iteration) %.0f"
          % (tag, sw.n_design * sw.n_case, pit / (sw.n_design * sw.n_case), ms, 1e6 * ms / pit, v.sum() / pit))
    for nm, x in zip(names, v):
        if x:
            print("    %-16s %9.0f" % (nm, x / pit))
    ctx.close()
    return v / pit


w = None
ctx0 = lib.context(0)
sw0, _, _ = bench.make_sweep(ctx0, 8, 0, pinned=False)
w = sw0.w
ctx0.close()
dw = float(w[1] - w[0])
sea = lambda Hs, Tp: np.sqrt(2.0 * waves.jonswap(w, Hs, Tp) * dw)
a = run("A one sea state")
b = run("B three sea states", zeta=np.stack([sea(6.0, 12.0), sea(4.0, 10.0), sea(8.0, 14.0)])[:, None, :], beta=np.zeros((3, 1)))
c = run("C the same sea state three times", zeta=np.stack([sea(6.0, 12.0)] * 3)[:, None, :], beta=np.zeros((3, 1)))
print("ratio C / A per phase:", {nm: round(float(y / x), 3) for nm, x, y in zip(names, a, c) if x})
