"""TEST INFRASTRUCTURE ONLY -- times the UNMODIFIED reference (raft.Model.solveDynamics,
raft/raft_model.py:966) on this container's host cores, on a sample of the C3 sweep, and
writes profiles/reference_cpu_timing.json.  Only runs where /root/reference exists (not on
the GPU box: bench.py's cpu_baseline there is the compiled oracle, kind "port").

usage: python oracle/time_reference.py [n_designs=4]
"""
import copy
import json
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
from oracle import ref_harness as rh          # noqa: E402
from oracle.make_golden import volturnus_variant, REF   # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    base = rh.prepare_design(rh.load_design(os.path.join(REF, "examples/VolturnUS-S_example.yaml")))
    scales = np.random.default_rng(0).uniform(0.75, 1.25, size=(64, 5))
    case = rh.make_case(Hs=6.0, Tp=12.0, heading=0.0)
    t_build = t_solve = 0.0
    nw = None
    for i in range(n):
        t0 = time.perf_counter()
        m = rh.build_model(volturnus_variant(base, scales[i]))
        t1 = time.perf_counter()
        m.solveDynamics(copy.deepcopy(case))
        t2 = time.perf_counter()
        t_build += t1 - t0
        t_solve += t2 - t1
        nw = m.nw
    out = {"what": "raft.Model.solveDynamics (reference, NumPy), C3 sweep variants, 1 sea state, %d bins" % nw,
           "designs": n, "cores": 1, "host": "build container (%d logical cores)" % (os.cpu_count() or 0),
           "solveDynamics_s_per_design": t_solve / n, "model_build_s_per_design": t_build / n,
           "dcf_per_s_per_core": n * nw / t_solve}
    os.makedirs(os.path.join(ROOT, "profiles"), exist_ok=True)
    with open(os.path.join(ROOT, "profiles", "reference_cpu_timing.json"), "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
