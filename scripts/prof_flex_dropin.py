"""Where one drop-in Model.solveDynamics call of the reference's flexible deck goes (host profile on the GPU box)."""
import cProfile, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raft_amd import backend, dropin
from raft_amd.snapshot import load_model_fixture, case_from_fixture
fx, model = load_model_fixture("flex_volturnus.npz")
ctx = backend.default_context(0)
eng = dropin.Engine(ctx)
base = case_from_fixture(fx["cases"][0])
for _ in range(3):
    t0 = time.perf_counter(); eng.solveDynamics(model, dict(base)); print("call %.3f ms, fixed point on the device %.3f ms" % (1e3 * (time.perf_counter() - t0), ctx.last_kernel_ms()))
pr = cProfile.Profile(); pr.enable()
eng.solveDynamics(model, dict(base))
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(22)
