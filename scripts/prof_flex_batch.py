"""FlexSweep.run over repeated calls on the GPU box: wall per call, and the host profile of the slow ones."""
import cProfile, io, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from raft_amd import backend, dropin
from raft_amd.snapshot import load_model_fixture, case_from_fixture
fx, model = load_model_fixture("flex_volturnus.npz")
ctx = backend.default_context(0)
eng = dropin.Engine(ctx)
base = case_from_fixture(fx["cases"][0])
cases = [base, dict(base, wave_height=4.0, wave_period=9.0, wave_heading=-20.0), dict(base, wave_height=1.0, wave_period=6.0)]
sw = dropin.flex_sweep_from_models([model] * 16, cases)
tot, shown = [], 0
for rep in range(14):
    if rep % 3 == 0:
        eng.solveDynamics(model, dict(cases[rep % 2]))
    time.sleep(float(os.environ.get('IDLE_MS', '0')) * 1e-3)
    pr = cProfile.Profile()
    t0 = time.perf_counter(); pr.enable(); sw.run(ctx); pr.disable(); tot.append(1e3 * (time.perf_counter() - t0))
    if tot[-1] > 38 and rep > 1 and shown < 2:
        shown += 1
        b = io.StringIO(); pstats.Stats(pr, stream=b).sort_stats("tottime").print_stats(6); print("call %d: %.1f ms" % (rep, tot[-1])); print("\n".join(b.getvalue().split("\n")[6:16]))
print("total   ", " ".join("%6.1f" % t for t in tot))
