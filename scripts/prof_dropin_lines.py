"""Line-level wall time of Engine.solveDynamics (GPU box): where a drop-in call goes.  argv[1]: fixture (default the C2 deck;
`flex_volturnus.npz` for the 150-DOF flexible deck)."""
import collections, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tests.util import load_model_fixture, case_from_fixture
from raft_amd import dropin, backend
ctx = backend.default_context(0)
fx, m = load_model_fixture(sys.argv[1] if len(sys.argv) > 1 else "c2_volturnus.npz")
eng = dropin.Engine(ctx)
case = case_from_fixture(fx["cases"][0])
for _ in range(5):
    eng.solveDynamics(m, dict(case))
t0 = time.perf_counter()
for _ in range(50):
    eng.solveDynamics(m, dict(case))
print("ms/call untraced: %.3f" % ((time.perf_counter() - t0) / 50 * 1e3))
acc = collections.Counter(); last = [None, None]
codes = {dropin.Engine.solveDynamics.__code__: "solveDynamics", dropin.Engine._upload.__code__: "_upload",
         dropin.Engine._bem_excitation_units.__code__: "_bem", dropin.Engine._solve_general.__code__: "_solve_general",
         dropin.Engine._excitation_general.__code__: "_excitation_general", dropin.Engine._node_units.__code__: "_node_units"}
def tr(frame, event, arg):
    if frame.f_code in codes:
        name = codes[frame.f_code]
        def local(frame, event, arg):
            now = time.perf_counter()
            if last[0] is not None: acc[last[0]] += now - last[1]
            last[0] = (name, frame.f_lineno); last[1] = now
            return local
        return local
    return None
sys.settrace(tr)
for _ in range(30): eng.solveDynamics(m, dict(case))
sys.settrace(None)
src = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "raft_amd", "dropin.py")).read().split("\n")
for (nm, ln), t in acc.most_common(18):
    print("%7.3f ms  %s L%d: %s" % (t / 30 * 1e3, nm, ln, src[ln - 1].strip()[:100]))
