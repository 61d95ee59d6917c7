#!/usr/bin/env python
"""bench.py -- design-case-frequency (dcf) solves / second on MI355X.

Workload (BASELINE.json configs[2], the one the north-star target is quoted
on): a VolturnUS-S geometry sweep, nDesign designs x 1 sea state (JONSWAP
Hs 6 m, Tp 12 s, head seas) x 200 frequency bins, nIter=4 (5 fixed-point
iterations), all fp64.  The designs are nDesign DISTINCT variants of the five
parameters of raft/parametersweep.py:33-37, each x U[0.75,1.25]
(default_rng(0)), exactly as SURVEY.md 8d defines C3: their member
descriptions are edited on the host (vectorised NumPy, tests/util.py) and the
strip tables, Morison added mass, hydrostatics and member inertia are GENERATED
ON THE DEVICE (raftx_build_designs) before the timed region.  The first 64
variants are the ones the live reference built for tests/golden/c3_variants.npz,
so rank 0 checks its responses against the reference's own solveDynamics.
A "step" is one pass of the whole hot path (raftx_solve_dynamics_device: strip
sweep + drag-linearisation fixed point + per-bin 6x6 complex solves) over every
design of this rank; inputs are resident in HBM before the timed region starts
and the responses stay in HBM.  (--tiled: the older workload, the 64
reference-built strip tables tiled round-robin and uploaded.)

Multi-GPU: one process per GPU (torch.distributed / RCCL is plumbing only:
barrier + max-over-ranks of the timing + result checksums).  Designs are
independent, so ranks shard them with no data-path collective -> weak scaling
(nDesign per rank is fixed).

Prints ONE JSON line (rank 0).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8 TB/s spec
FP64_VALU_PEAK_TF = 78.6       # MI355X vector fp64 peak


def load_sweep(n_design, rank=0):
    """Strip tables / matrices of the sweep.  The GPU box has no reference
    tree, so the designs come from the committed C3 sample
    (tests/golden/c3_variants.npz: 64 true parametersweep variants built by
    the live reference) tiled round-robin, each rank starting at a different
    offset."""
    from tests import standin
    fx = standin.load_fixture("c3_variants.npz")
    off = fx["strip_offsets"]
    nV = len(off) - 1
    idx = (np.arange(n_design) + rank * 7) % nV
    counts = (off[1:] - off[:-1])[idx]
    new_off = np.concatenate([[0], np.cumsum(counts)]).astype(np.int64)
    strips = np.concatenate([fx["strips"][off[i]:off[i + 1]] for i in idx], axis=0)
    return dict(off=new_off, strips=strips, M0=fx["M0"][idx], B0=fx["B0"][idx], C0=fx["C0"][idx],
                w=fx["w"], k=fx["k"], depth=fx["depth"], zeta=fx["zeta"], beta=fx["beta"],
                nIter=int(fx["nIter"]), XiStart=float(fx["XiStart"]), idx=idx, fx=fx)


def generate_sweep(ctx, n_design, rank=0):
    """The C3 sweep generated on the device: descriptors (host, vectorised) -> raftx_build_designs.  Rank r takes
    rows [r*n, (r+1)*n) of one default_rng(0) draw, so rank 0's first 64 designs are the committed reference-built
    variants.  Returns the dict the rest of this file uses (offsets, matrices, sea state, timings)."""
    from tests import standin
    from tests.util import volturnus_sweep
    from raft_amd import geometry as G
    fx = standin.load_fixture("c3_variants.npz")
    fg = standin.load_fixture("geom_units.npz")
    base = json.loads(fg["c3_base_json"])
    u0 = [u for u in fg["units"] if u["name"] == "C3-variant-0"][0]
    # constants that are not geometry: rotor-nacelle assembly (live reference minus its massless-RNA twin), mooring
    M_rna = np.asarray(u0["M_struc"]) - np.asarray(u0["M_struc_bare"])
    C_rest = np.asarray(u0["C_struc"]) - np.asarray(u0["C_struc_bare"]) + np.diag([7e4, 7e4, 0.0, 0.0, 0.0, 1e8])
    scales = np.random.default_rng(0).uniform(0.75, 1.25, size=((rank + 1) * n_design, 5))[rank * n_design:]
    nw = len(fx["w"])
    t0 = time.perf_counter()
    D = volturnus_sweep(base, scales).tables()
    t_desc = time.perf_counter() - t0
    M0 = np.repeat(M_rna[None], n_design, axis=0)
    B0 = np.repeat(np.asarray(fx["B0"])[:1], n_design, axis=0)
    C0 = np.repeat(C_rest[None], n_design, axis=0)

    def build():
        return ctx.build_designs(D.member_off, D.members, D.station_off, D.stations, M0, B0, C0, nw, rho=1025.0, g=9.81,
                                 cap_off=D.cap_off, caps=D.caps, add_mask=G.ADD_MORISON | G.ADD_HYDROSTATIC | G.ADD_INERTIA)
    build()                                   # first call pays the allocator; time the second
    t0 = time.perf_counter()
    off = build()
    t_build = time.perf_counter() - t0
    geo = {"designs": int(n_design), "members": int(D.member_off[-1]), "strips": int(off[-1]),
           "host_descriptor_ms": 1e3 * t_desc, "build_designs_wall_ms": 1e3 * t_build,
           "build_designs_kernels_ms": ctx.last_kernel_ms(),
           "descriptor_bytes": int(D.members.nbytes + D.stations.nbytes + D.caps.nbytes),
           "strip_table_bytes_not_uploaded": int(off[-1]) * 256}
    return dict(off=off, strips=None, M_extra=M0, C_extra=C0, B0=B0, w=fx["w"], k=fx["k"], depth=fx["depth"], zeta=fx["zeta"], beta=fx["beta"],
                nIter=int(fx["nIter"]), XiStart=float(fx["XiStart"]), idx=np.arange(n_design) if rank == 0 else np.full(n_design, -1),
                fx=fx, rebuild=build, geometry=geo)


def measured_traffic(n_design):
    """HBM-side bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes
    (profiles/traffic_latest.json, written by scripts/gpu_traffic.sh + scripts/traffic_summary.py;
    FETCH_SIZE and WRITE_SIZE are collected in separate passes and FETCH_SIZE is doubled as
    MI355X_MICROARCH.md prescribes for gfx950).  None if no profile matches this workload."""
    path = os.path.join(ROOT, "profiles", "traffic_latest.json")
    if not os.path.exists(path):
        return None
    with open(path) as f:
        t = json.load(f)
    if int(t.get("designs_per_gpu", -1)) != int(n_design):
        return None
    return float(t["hbm_bytes_per_launch"])


def algorithmic_bytes(sw):
    """A_min = 256*S + 3*288 + 8*nw + 96*nw per (design, case)."""
    nw = len(sw["w"])
    S = (sw["off"][1:] - sw["off"][:-1]).astype(np.float64)
    return float(np.sum(256.0 * S + 864.0 + 104.0 * nw))


def algorithmic_flops(sw, niter):
    """N_it*(175*S + 2000) + 160*S fp64 flops per dcf (transcendentals not counted)."""
    nw = len(sw["w"])
    S = (sw["off"][1:] - sw["off"][:-1]).astype(np.float64)
    return float(np.sum(niter.reshape(-1) * (175.0 * S + 2000.0) + 160.0 * S) * nw)


def cpu_baseline(sw, seconds_target=12.0, ctx=None):
    """The oracle (oracle/raftx_oracle.c, kind="port") timed on this host's
    cores on a bounded sample of the same workload."""
    import subprocess
    from raft_amd._abi import RaftxLib
    so = os.path.join(ROOT, "oracle", "libraftx_oracle.so")
    if not os.path.exists(so):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")], stdout=subprocess.DEVNULL)
    lib = RaftxLib(so)
    lib.lib.raftx_oracle_threads.restype = int
    threads = int(lib.lib.raftx_oracle_threads())
    nw = len(sw["w"])
    if sw.get("strips") is None:              # device-generated sweep: the oracle gets the same tables, fetched once
        strips, _ = ctx.fetch_strips(sw["off"][-1])
        S = ctx.fetch_statics()
        sw = dict(sw, strips=strips, M0=S["M_struc"] + S["A_morison"] + sw["M_extra"], C0=S["C_struc"] + S["C_hydro"] + sw["C_extra"])

    def run(n):
        ctx = lib.context(0)
        ctx.upload_designs_raw(sw["off"][:n + 1], sw["strips"][:sw["off"][n]], sw["M0"][:n], sw["B0"][:n],
                               sw["C0"][:n], nw)
        ctx.upload_cases(sw["w"], sw["k"], sw["depth"], 1025.0, 9.81, sw["zeta"][None], sw["beta"][None])
        t0 = time.perf_counter()
        ctx.solve_dynamics_device(sw["nIter"], 0.01, sw["XiStart"])
        dt = time.perf_counter() - t0
        ctx.close()
        return dt

    n0 = min(len(sw["off"]) - 1, max(threads, 8))
    dt0 = run(n0)
    n = int(min(len(sw["off"]) - 1, max(n0, n0 * seconds_target / max(dt0, 1e-3))))
    dt = run(n)
    return {"value": n * nw / dt, "unit": "dcf solves/s", "cores": threads, "kind": "port",
            "sample": "%d of the sweep's designs x 1 sea state x %d bins, oracle/raftx_oracle.c (OpenMP, %d threads), %.1f s"
                      % (n, nw, threads, dt)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--designs", type=int, default=10000, help="designs per GPU (weak scaling)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--tiled", action="store_true", help="older workload: the 64 reference-built strip tables tiled and uploaded")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    # RAFTX_BENCH_BACKEND=gloo + RAFTX_BENCH_DEVICE=0: rehearsal of the multi-rank path on a single-GPU box (all ranks
    # share device 0, host tensors for the two collectives); the driver's runs use RCCL with one GPU per rank
    backend_name = os.environ.get("RAFTX_BENCH_BACKEND", "nccl")
    if "RAFTX_BENCH_DEVICE" in os.environ:
        local = int(os.environ["RAFTX_BENCH_DEVICE"])
    if world > 1:
        import torch
        import torch.distributed as dist
        torch.cuda.set_device(local)
        if backend_name == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend=backend_name)

    from raft_amd import backend
    ctx = backend.hip_library().context(local)

    if args.tiled:
        sw = load_sweep(args.designs, rank)
        nw = len(sw["w"])
        ctx.upload_designs_raw(sw["off"], sw["strips"], sw["M0"], sw["B0"], sw["C0"], nw)
    else:
        sw = generate_sweep(ctx, args.designs, rank)
        nw = len(sw["w"])
    ctx.upload_cases(sw["w"], sw["k"], sw["depth"], 1025.0, 9.81, sw["zeta"][None], sw["beta"][None])

    def barrier():
        if dist is not None:
            import torch
            torch.cuda.synchronize()
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        ctx.solve_dynamics_device(sw["nIter"], 0.01, sw["XiStart"])
    barrier()
    t0 = time.perf_counter()
    kern_ms = []
    for _ in range(args.steps):
        ctx.solve_dynamics_device(sw["nIter"], 0.01, sw["XiStart"])      # synchronous: returns after the stream drains
        kern_ms.append(ctx.last_kernel_ms())                             # HIP events on the ctx stream
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        import torch
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if backend_name == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # correctness spot check of what was just timed (outside the timed region)
    res = ctx.fetch_results(want_Xi=True)
    niter = res["niter"]
    nan = int(np.count_nonzero(res["flags"] & 2))
    from tests.util import rao_group_err
    errs = []
    for j, sol in enumerate(sw["fx"]["solved"]):
        hits = np.nonzero(sw["idx"] == j)[0][:1]
        if len(hits):                 # SURVEY.md 8d metric: group-relative error of the RAOs (Xi / zeta)
            errs.append(rao_group_err(res["Xi"][hits[0], 0, 0], sol["Xi"][0], sw["zeta"][0]))
            assert int(niter[hits[0], 0]) == int(sol["units"][0]["niter"]), "iteration count differs from the reference"
    max_err = float(max(errs)) if errs else None
    assert nan == 0 and (max_err is None or max_err < 1e-6), "bench results fail parity (err=%r, nan=%d)" % (max_err, nan)

    # PCIe-inclusive rate of one whole boundary crossing (H2D of the tables + launch + D2H of Xi): informational,
    # never `value`
    Xi_pinned = ctx.pinned_empty(res["Xi"].shape)        # page-locked landing buffer for the responses (raftx_host_alloc)
    t0 = time.perf_counter()
    if args.tiled:
        ctx.upload_designs_raw(sw["off"], sw["strips"], sw["M0"], sw["B0"], sw["C0"], nw)
    else:
        sw["rebuild"]()                         # descriptor H2D + device generation
    ctx.upload_cases(sw["w"], sw["k"], sw["depth"], 1025.0, 9.81, sw["zeta"][None], sw["beta"][None])
    ctx.solve_dynamics_device(sw["nIter"], 0.01, sw["XiStart"])
    ctx.fetch_results(Xi_out=Xi_pinned)
    pcie_rate = args.designs * nw / (time.perf_counter() - t0)
    assert np.array_equal(Xi_pinned.view(np.uint64), res["Xi"].view(np.uint64))
    ctx.free_pinned(Xi_pinned)
    # ... and of the optimiser-style crossing: descriptors in, response statistics out (no 19 KB/design-case download)
    t0 = time.perf_counter()
    if args.tiled:
        ctx.upload_designs_raw(sw["off"], sw["strips"], sw["M0"], sw["B0"], sw["C0"], nw)
    else:
        sw["rebuild"]()
    ctx.upload_cases(sw["w"], sw["k"], sw["depth"], 1025.0, 9.81, sw["zeta"][None], sw["beta"][None])
    ctx.solve_dynamics_device(sw["nIter"], 0.01, sw["XiStart"])
    ctx.motion_stats(float(sw["w"][1] - sw["w"][0]))
    ctx.fetch_results(want_Xi=False)
    stats_rate = args.designs * nw / (time.perf_counter() - t0)

    n_dcf_rank = args.designs * 1 * nw
    total_dcf = n_dcf_rank * world * args.steps
    value = total_dcf / elapsed
    k_ms = float(np.mean(kern_ms))
    A = algorithmic_bytes(sw)
    flops = algorithmic_flops(sw, niter)
    out = {
        "metric": "design-case-frequency solves/sec (whole node)",
        "value": value, "unit": "dcf solves/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1e3 * elapsed / args.steps,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f64", "data": "synthetic",
        "config": {"workload": "C3 VolturnUS-S parameter sweep: %d designs/GPU x 1 sea state (JONSWAP Hs6 Tp12, 0 deg) x %d bins, "
                               "nIter=4, tol=0.01; %s" % (args.designs, nw, "designs = 64 reference-built sweep variants tiled" if args.tiled else
                                                          "designs = distinct U[0.75,1.25]^5 variants (default_rng(0)), generated on the device"),
                   "designs_per_gpu": args.designs, "cases": 1, "nw": nw, "sharding": "designs over ranks, no collective",
                   "designs_from": "tiled reference-built strip tables (upload)" if args.tiled else
                                   "distinct variants, strip tables + statics generated on the device (raftx_build_designs)"},
        "rao_max_rel_err_vs_reference": max_err,
        "pcie_inclusive_dcf_per_s_per_gpu": pcie_rate,
        "pcie_inclusive_stats_only_dcf_per_s_per_gpu": stats_rate,
        "mean_iterations": float(np.mean(niter)),
        "roofline": {"bound": "hbm", "achieved": A / (k_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": A / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": measured_traffic(args.designs),
                     "kernel": "k_solve_dynamics", "kernel_ms": k_ms, "algorithmic_bytes_per_launch": A,
                     "note": "fused kernel is fp64-VALU-bound (SURVEY.md 8d): see roofline_fp64_valu"},
        "roofline_fp64_valu": {"achieved": flops / (k_ms * 1e-3) / 1e12, "peak": FP64_VALU_PEAK_TF, "unit": "TFLOP/s",
                               "frac": flops / (k_ms * 1e-3) / 1e12 / FP64_VALU_PEAK_TF,
                               "algorithmic_flops_per_launch": flops},
    }
    ref_path = os.path.join(ROOT, "profiles", "reference_cpu_timing.json")
    if os.path.exists(ref_path):      # the unmodified NumPy reference, timed in the BUILD container (it cannot travel to the GPU box)
        with open(ref_path) as f:
            rt = json.load(f)
        out["reference_numpy_build_container"] = {"dcf_per_s_one_core": rt.get("dcf_per_s_per_core"),
                                                  "dcf_per_s_pool": rt.get("pool", {}).get("dcf_per_s_all_cores_solve_only"),
                                                  "pool_cores": rt.get("pool", {}).get("cores")}
    if not args.tiled:
        out["geometry"] = sw["geometry"]
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(sw, ctx=ctx)
    if rank == 0:
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
