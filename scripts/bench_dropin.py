#!/usr/bin/env python
"""Wall time of ONE drop-in Model.solveDynamics call (raft_amd.dropin) on the GPU, for C1 (OC3spar, nw = 50) and C2
(VolturnUS-S, nw = 200, three sea states): host packing of the strip table from the Member objects, upload, the fused
kernel, coupled solve and downloads -- everything a caller of the reference's method waits for.  Beside it: the
reference's own NumPy solveDynamics on the same cases (seconds recorded by oracle/make_golden.py when the goldens were
generated in the build container; the reference tree does not travel to the GPU box).

    python scripts/bench_dropin.py [--repeat 7] > profiles/r02_dropin.json
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from raft_amd import backend, dropin, snapshot                       # noqa: E402
from raft_amd.metrics import group_rel_err                           # noqa: E402
from raft_amd import strips as strips_mod                            # noqa: E402


def case_dict(c):
    return {k: (list(v) if isinstance(v, (list, np.ndarray)) else v) for k, v in c["case"].items()}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--repeat", type=int, default=7)
    args = ap.parse_args()
    eng = dropin.Engine(backend.default_context(0))
    out = {"what": "one drop-in Model.solveDynamics call on MI355X (stand-in Model/FOWT/Member objects rebuilt from the "
                   "live-reference snapshots under tests/golden/)", "repeat": args.repeat, "configs": []}
    for name, fixture in (("C1 OC3spar nw=50", "c1_oc3spar.npz"), ("C2 VolturnUS-S nw=200", "c2_volturnus.npz"),
                          ("VolturnUS-S-flexible, 150 reduced DOFs, nw=40 (node-by-node sweeps + raftx_solve_dense)",
                           "flex_volturnus.npz")):
        fx = snapshot.load_fixture(fixture)
        model = snapshot.build_model(fx["model"])
        rows = []
        for c in fx["cases"]:
            case = case_dict(c)
            t_call, t_pack = [], []
            Xi = None
            for _ in range(args.repeat + 1):                       # first call warms the context (allocations, module load)
                t0 = time.perf_counter()
                Xi = eng.solveDynamics(model, dict(case)).copy()
                t_call.append(time.perf_counter() - t0)
                t0 = time.perf_counter()
                if model.fowtList[0].nDOF == 6:
                    strips_mod.pack_fowt(model.fowtList[0])
                else:
                    strips_mod.pack_fowt_nodes(model.fowtList[0])
                t_pack.append(time.perf_counter() - t0)
            nH = Xi.shape[0] - 1
            ref = np.asarray(c["Xi"])[:nH]
            err = group_rel_err(Xi[:nH], ref) if model.fowtList[0].nDOF == 6 else np.abs(Xi[:nH] - ref).max() / np.abs(ref).max()
            rows.append({"wave": [case.get("wave_height"), case.get("wave_period"), case.get("wave_heading")],
                         "gpu_call_ms_median": 1e3 * float(np.median(t_call[1:])), "gpu_call_ms_min": 1e3 * float(np.min(t_call[1:])),
                         "of_which_host_strip_packing_ms": 1e3 * float(np.median(t_pack[1:])),
                         "first_call_ms": 1e3 * t_call[0],
                         "reference_numpy_s": float(c["ref_seconds"]), "niter": int(model._raftx_niter[0]),
                         "rel_err_vs_reference": float(err),
                         "speedup_vs_reference": float(c["ref_seconds"]) / float(np.median(t_call[1:]))})
        out["configs"].append({"config": name, "nw": int(model.nw), "cases": rows})
    out["note"] = ("reference_numpy_s: the unmodified reference's solveDynamics on one core of the build container, recorded "
                   "when the golden was generated; gpu_call_ms: Python call to return, kernels + copies + host packing")
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
