"""TEST / BASELINE INFRASTRUCTURE ONLY -- times the UNMODIFIED reference (raft.Model.solveDynamics,
raft/raft_model.py:966) on THIS host's cores, on a bounded sample of the C3 sweep (VolturnUS-S_example,
the five parametersweep.py parameters x U[0.75,1.25], default_rng(0); 1 sea state Hs 6 m / Tp 12 s; 200 bins), under the
stub recipe of SURVEY.md 8c (oracle/ref_harness.py: no MoorPy system, aeroServoMod 0, injected C_moor).

The reference is imported from /root/reference where that tree exists (the build container) or from the byte-compiled
archive oracle/_ref/raft_reference.zip (built by oracle/stage_reference.py; what the GPU box has).  Every timed solve
is checked against the committed live-reference fixture tests/golden/c3_variants.npz (same designs, same case): the
thing timed is the thing the parity tests are pinned on.

usage: python oracle/time_reference.py [--designs N]                 one process, one core (BLAS threads = 1)
       python oracle/time_reference.py --pool [--procs P] [--items M] multiprocessing.Pool(P) over M (design, case) items
                                                                     (SURVEY.md 8d (ii)); BLAS / OpenMP threads = 1
       add --write to update profiles/reference_cpu_timing.json (build container only)
Prints ONE JSON line.
"""
import argparse
import contextlib
import copy
import io
import json
import os
import sys
import time

for _k in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS"):      # before numpy is imported, workers inherit
    os.environ[_k] = "1"

import numpy as np                              # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
from oracle import ref_harness as rh            # noqa: E402

DECK = "examples/VolturnUS-S_example.yaml"
N_FIXTURE = 64                                   # variants of tests/golden/c3_variants.npz with a stored reference solve


def usable_cores():
    try:
        return len(os.sched_getaffinity(0))
    except AttributeError:
        return os.cpu_count() or 1


def _spin(seconds):
    t_end = time.perf_counter() + seconds
    c0 = time.process_time()
    x = 0
    while time.perf_counter() < t_end:
        x += 1
    return time.process_time() - c0


def cgroup_cpu_limit():
    """CPUs the container's CFS quota allows (cgroup v2 cpu.max / v1 cpu.cfs_quota_us), or None when unlimited / unknown."""
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            q, p = f.read().split()[:2]
        if q != "max":
            return float(q) / float(p)
    except (OSError, ValueError):
        pass
    try:
        with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as f:
            q = float(f.read())
        with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f:
            p = float(f.read())
        if q > 0:
            return q / p
    except (OSError, ValueError):
        pass
    return None


def probe_parallelism(seconds=1.0):
    """How many of this host's logical CPUs the container may actually RUN on at once: one busy loop per logical CPU for
    ``seconds`` of wall time, CPU time summed / wall time.  (The GPU boxes of the pool show 256 logical CPUs to a container
    whose CPU-time quota is a small fraction of that: a Pool(256) there is 256 processes taking turns.)"""
    import multiprocessing as mp
    n = usable_cores()
    ctx = mp.get_context("fork")
    with ctx.Pool(n) as pool:
        pool.map(_spin, [0.05] * n, chunksize=1)             # every worker alive before the measured second
        cpu = pool.map(_spin, [seconds] * n, chunksize=1)
    eff = sum(cpu) / seconds
    return {"logical_cpus": n, "effective_parallel_cpus": eff, "cgroup_cpu_limit": cgroup_cpu_limit(),
            "mean_share_per_process": eff / n}


def _variant(design, scales):
    from oracle.make_golden import volturnus_variant
    return volturnus_variant(design, scales)


_STATE = {}


def _setup():
    if not _STATE:
        from raft_amd import snapshot as standin
        fx = standin.load_fixture("c3_variants.npz")
        _STATE["scales"] = np.asarray(fx["scales"])
        _STATE["gold"] = fx["solved"]
        _STATE["base"] = rh.prepare_design(rh.load_deck(DECK))
        _STATE["case"] = rh.make_case(Hs=6.0, Tp=12.0, heading=0.0)
        rh.import_raft()
    return _STATE


def _one(i):
    """Build + solve variant (i mod 64); returns (build s, solve s, nw, max |Xi - golden| / max |golden|)."""
    st = _setup()
    j = i % N_FIXTURE
    with contextlib.redirect_stdout(io.StringIO()):
        t0 = time.perf_counter()
        m = rh.build_model(_variant(st["base"], st["scales"][j]))
        t1 = time.perf_counter()
        Xi = m.solveDynamics(copy.deepcopy(st["case"]))
        t2 = time.perf_counter()
    ref = np.asarray(st["gold"][j]["Xi"])
    got = np.asarray(Xi)[:ref.shape[0]]
    err = float(np.max(np.abs(got - ref)) / np.max(np.abs(ref)))
    return t1 - t0, t2 - t1, int(m.nw), err


def run_single(n):
    res = [_one(i) for i in range(n)]
    nw = res[0][2]
    t_build = sum(r[0] for r in res)
    t_solve = sum(r[1] for r in res)
    return {"mode": "single", "what": "raft.Model.solveDynamics (unmodified reference, NumPy/SciPy), C3 sweep variants, "
            "1 sea state, %d bins" % nw, "designs": n, "cores": 1, "cores_on_host": usable_cores(),
            "solveDynamics_s_per_design": t_solve / n, "model_build_s_per_design": t_build / n,
            "dcf_per_s_one_core": n * nw / t_solve, "dcf_per_s_one_core_incl_model_build": n * nw / (t_solve + t_build),
            "max_rel_err_vs_committed_reference_fixture": max(r[3] for r in res), "reference_from": rh.reference_kind()}


def run_pool(procs, items):
    import multiprocessing as mp
    ctx = mp.get_context("fork")
    with ctx.Pool(procs) as pool:
        pool.map(_warm, range(procs), chunksize=1)           # imports + fixture in every worker, outside the timed region
        t0 = time.perf_counter()
        res = pool.map(_one, range(items), chunksize=1)
        wall = time.perf_counter() - t0
    nw = res[0][2]
    busy = sum(r[0] + r[1] for r in res)
    return {"mode": "pool", "what": "multiprocessing.Pool(%d) over %d (design, case) items of the C3 sweep, BLAS/OpenMP threads = 1; "
            "each item = Model(design) + statics + hydro constants (untimed part of a solve in the reference's own "
            "analyzeCases) + solveDynamics" % (procs, items), "procs": procs, "items": items, "cores_on_host": usable_cores(),
            "wall_s": wall, "dcf_per_s_pool": items * nw / wall,
            "dcf_per_s_pool_solve_only": items * nw / (wall * sum(r[1] for r in res) / busy),
            "mean_solveDynamics_s": sum(r[1] for r in res) / items, "mean_model_build_s": sum(r[0] for r in res) / items,
            "worker_busy_fraction": busy / (wall * procs),
            "max_rel_err_vs_committed_reference_fixture": max(r[3] for r in res), "reference_from": rh.reference_kind()}


def _warm(_):
    _setup()
    time.sleep(0.05)                                         # let every worker take one warm-up item
    return 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--pool", action="store_true")
    ap.add_argument("--designs", type=int, default=2)
    ap.add_argument("--procs", type=int, default=0)
    ap.add_argument("--items", type=int, default=0)
    ap.add_argument("--write", action="store_true")
    ap.add_argument("--probe", action="store_true", help="only measure how many CPUs this container can run on at once")
    a = ap.parse_args()
    if a.probe:
        print(json.dumps(probe_parallelism()))
        return
    if a.pool:
        procs = a.procs or usable_cores()
        out = run_pool(procs, a.items or 2 * procs)
    else:
        out = run_single(a.designs)
    if a.write:
        path = os.path.join(ROOT, "profiles", "reference_cpu_timing.json")
        cur = json.load(open(path)) if os.path.exists(path) else {}
        cur[out["mode"]] = out
        json.dump(cur, open(path, "w"), indent=1)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
