"""TEST INFRASTRUCTURE ONLY -- times the UNMODIFIED reference (raft.Model.solveDynamics,
raft/raft_model.py:966) on this container's host cores, on a sample of the C3 sweep, and
writes profiles/reference_cpu_timing.json.  Only runs where /root/reference exists (not on
the GPU box: bench.py's cpu_baseline there is the compiled oracle, kind "port").

usage: python oracle/time_reference.py [n_designs=4]          single process, one core
       python oracle/time_reference.py --pool [n_designs=16] multiprocessing.Pool over all cores,
                                                              OMP/OPENBLAS threads pinned to 1
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


def _one(i):
    """Worker of the pool run: build + solve one C3 variant, return (solve seconds, nw)."""
    import io, contextlib
    base = rh.prepare_design(rh.load_design(os.path.join(REF, "examples/VolturnUS-S_example.yaml")))
    scales = np.random.default_rng(0).uniform(0.75, 1.25, size=(64, 5))
    case = rh.make_case(Hs=6.0, Tp=12.0, heading=0.0)
    with contextlib.redirect_stdout(io.StringIO()):
        m = rh.build_model(volturnus_variant(base, scales[i % 64]))
        t0 = time.perf_counter()
        m.solveDynamics(copy.deepcopy(case))
    return time.perf_counter() - t0, m.nw


def pool_main(n):
    import multiprocessing as mp
    for k_ in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS"):
        os.environ[k_] = "1"
    cores = os.cpu_count() or 1
    t0 = time.perf_counter()
    with mp.Pool(cores) as pool:
        res = pool.map(_one, range(n))
    wall = time.perf_counter() - t0
    nw = res[0][1]
    path = os.path.join(ROOT, "profiles", "reference_cpu_timing.json")
    out = json.load(open(path)) if os.path.exists(path) else {}
    out["pool"] = {"designs": n, "cores": cores, "wall_s_incl_model_build": wall,
                   "dcf_per_s_all_cores_incl_model_build": n * nw / wall,
                   "dcf_per_s_all_cores_solve_only": n * nw / (sum(r[0] for r in res) / cores),
                   "note": "multiprocessing.Pool(%d) over designs, BLAS threads = 1" % cores}
    json.dump(out, open(path, "w"), indent=1)
    print(json.dumps(out["pool"]))


def main():
    if "--pool" in sys.argv:
        rest = [a for a in sys.argv[1:] if a != "--pool"]
        return pool_main(int(rest[0]) if rest else 16)
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
           "designs": n, "cores": 1, "host": "%d logical cores" % (os.cpu_count() or 0),
           "solveDynamics_s_per_design": t_solve / n, "model_build_s_per_design": t_build / n,
           "dcf_per_s_per_core": n * nw / t_solve}
    if not os.environ.get("RAFTX_REF_TIMING_NOWRITE"):        # bench.py's on-host leg only wants the line below
        os.makedirs(os.path.join(ROOT, "profiles"), exist_ok=True)
        with open(os.path.join(ROOT, "profiles", "reference_cpu_timing.json"), "w") as f:
            json.dump(out, f, indent=1)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
