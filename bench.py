#!/usr/bin/env python
"""bench.py -- design-case-frequency (dcf) solves / second on MI355X.

Workload (BASELINE.json configs[2], the one the north-star target is quoted
on; SURVEY.md 8d "C3"): a VolturnUS-S geometry sweep, nDesign DISTINCT variants
of the five parameters of raft/parametersweep.py:33-37, each x U[0.75,1.25]
(default_rng(0)), x 1 sea state (JONSWAP Hs 6 m, Tp 12 s, head seas) x 200
frequency bins, nIter=4 (up to 5 fixed-point iterations), all fp64.

A STEP is one whole pass of the solver stage as SURVEY.md 8d defines it --
"H2D of the tables + kernels + D2H": the member descriptions of every design
of this rank (host arrays, page-locked) go in, the response statistics
(std of the six motions + iteration counts + flags) come out, in ONE library
call (raftx_sweep_stats): descriptor H2D, strip-table / statics generation
(k_geom_*), the fused fixed point (k_solve_dynamics: strip sweeps +
drag-linearisation iterations + per-bin 6x6 complex solves), the statistics
kernel and the D2H, block-pipelined over internal streams.  `value` = dcf of all
ranks / wall time of the K timed steps.  Host work outside the step (editing the
descriptors of the variants: vectorised NumPy, reported as
geometry.host_descriptor_ms) is not part of the solver stage.  ("state": "stats
out"; --xi-out times the same step with the full responses downloaded too.)

Also on the JSON line:
  kernel_resident   the fused kernel alone on the whole batch with inputs and outputs
                    resident in HBM (what round 1 reported as `value`), only with --resident
                    (kept out of the default run so that the rocprofv3 average of
                    k_solve_dynamics describes one population of launches);
  roofline          dominant kernel k_solve_dynamics over the launches of the timed
                    region: algorithmic bytes (SURVEY.md 8d A_min) / summed HIP-event
                    durations vs HBM peak;  roofline_fp64_valu the same with the
                    algorithmic FLOPs vs the fp64 vector peak (the kernel is VALU-bound);
  parity            every design of the timed batch against the CPU oracle (RAO
                    group-relative error, iteration counts), the first 64 against the
                    live reference's own solveDynamics (tests/golden/c3_variants.npz);
  cpu_baseline      the oracle (oracle/raftx_oracle.c, kind "port") on this host.

Multi-GPU: one process per GPU; designs are block-partitioned over ranks, no
collective while solving; the statistics are gathered onto rank 0 INSIDE the
timed region (RCCL through the library's own communicator, raft_amd/comm.py;
torch.distributed is plumbing for barrier + max-over-ranks only).  Weak scaling:
nDesign per rank is fixed.

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
FP64_FMA_SUSTAINED_TF = 50.8      # measured: profiles/r02_valu_mfma_probe.jsonl (probe fma64)


def make_sweep(ctx, n_design, rank=0, pinned=True):
    """Descriptors of this rank's designs (host, vectorised): rank r takes rows [r*n, (r+1)*n) of one default_rng(0)
    draw, so rank 0's first 64 designs are the committed reference-built variants."""
    from raft_amd import snapshot
    from raft_amd import geometry as G
    from raft_amd.sweep import GeometrySweep
    fx = snapshot.load_fixture("c3_variants.npz")
    fg = snapshot.load_fixture("geom_units.npz")
    base = json.loads(fg["c3_base_json"])
    u0 = [u for u in fg["units"] if u["name"] == "C3-variant-0"][0]
    # constants that are not geometry: rotor-nacelle assembly (live reference minus its massless-RNA twin), mooring
    M_rna = np.asarray(u0["M_struc"]) - np.asarray(u0["M_struc_bare"])
    C_rest = np.asarray(u0["C_struc"]) - np.asarray(u0["C_struc_bare"]) + np.diag([7e4, 7e4, 0.0, 0.0, 0.0, 1e8])
    scales = np.random.default_rng(0).uniform(0.75, 1.25, size=((rank + 1) * n_design, 5))[rank * n_design:]
    t0 = time.perf_counter()
    D = G.volturnus_sweep(base, scales).tables()
    t_desc = time.perf_counter() - t0
    if pinned:                                          # page-locked staging (raftx_host_alloc): full-rate, asynchronous H2D
        for name in ("members", "stations", "caps", "member_off", "station_off", "cap_off"):
            a = getattr(D, name, None)
            if a is not None and a.size:
                b = ctx.pinned_empty(a.shape, dtype=a.dtype)
                b[...] = a
                setattr(D, name, b)
    M0 = np.repeat(M_rna[None], n_design, axis=0)
    B0 = np.repeat(np.asarray(fx["B0"])[:1], n_design, axis=0)
    C0 = np.repeat(C_rest[None], n_design, axis=0)
    if pinned:
        def _pin(a):
            b = ctx.pinned_empty(a.shape, dtype=np.float64)
            b[...] = a
            return b
        M0, B0, C0 = _pin(M0), _pin(B0), _pin(C0)
    sw = GeometrySweep(D, M0, B0, C0, fx["w"], fx["k"], float(fx["depth"]), np.asarray(fx["zeta"])[None], np.asarray(fx["beta"])[None],
                       int(fx["nIter"]), float(fx["XiStart"]), tol=0.01, add_mask=G.ADD_MORISON | G.ADD_HYDROSTATIC | G.ADD_INERTIA)
    geo = {"designs": int(n_design), "members": int(D.member_off[-1]), "host_descriptor_ms": 1e3 * t_desc,
           "descriptor_bytes": int(D.members.nbytes + D.stations.nbytes + D.caps.nbytes), "descriptors_page_locked": bool(pinned)}
    return sw, fx, geo


def measured_traffic(n_design):
    """HBM-side bytes of the dominant kernel from the committed rocprofv3 PMC passes (profiles/traffic_latest.json, written
    by scripts/gpu_traffic.sh + scripts/traffic_summary.py; FETCH_SIZE and WRITE_SIZE are collected in separate passes and
    FETCH_SIZE is doubled as MI355X_MICROARCH.md prescribes for gfx950), scaled to the designs of one step.  None if no
    profile exists."""
    path = os.path.join(ROOT, "profiles", "traffic_latest.json")
    if not os.path.exists(path):
        return None
    with open(path) as f:
        t = json.load(f)
    n_prof = float(t.get("designs_per_launch", t.get("designs_per_gpu", 0)) or 0)
    if n_prof <= 0:
        return None
    return float(t["hbm_bytes_per_launch"]) * float(n_design) / n_prof


def algorithmic_bytes(off, nw):
    """A_min = 256*S + 3*288 + 8*nw + 96*nw per (design, case)."""
    S = np.diff(off).astype(np.float64)
    return float(np.sum(256.0 * S + 864.0 + 104.0 * nw))


def algorithmic_flops(off, nw, niter):
    """N_it*(175*S + 2000) + 160*S fp64 flops per dcf (transcendentals not counted)."""
    S = np.diff(off).astype(np.float64)
    return float(np.sum(niter.reshape(-1) * (175.0 * S + 2000.0) + 160.0 * S) * nw)


def oracle_run(sw, ctx):
    """The CPU oracle (oracle/raftx_oracle.c) on this host, over EVERY design of the timed batch: its wall time is the
    cpu_baseline, its responses are the checker of the batch.  The oracle gets the strip tables and statics the device
    generated, fetched once."""
    import subprocess
    from raft_amd._abi import RaftxLib
    so = os.path.join(ROOT, "oracle", "libraftx_oracle_fast.so")          # same source as the checker, -O3 -march=x86-64-v3
    if not os.path.exists(so):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")], stdout=subprocess.DEVNULL)
    lib = RaftxLib(so)
    lib.lib.raftx_oracle_threads.restype = int
    threads = int(lib.lib.raftx_oracle_threads())
    nw = sw.nw
    sw.upload(ctx)                                        # device generation (outside every timed region)
    off = sw.off
    strips, _ = ctx.fetch_strips(off[-1])
    S = ctx.fetch_statics()
    M0 = S["M_struc"] + S["A_morison"] + sw.M0
    C0 = S["C_struc"] + S["C_hydro"] + sw.C0
    n = sw.n_design
    o = lib.context(0)
    o.upload_designs_raw(off, strips, M0, sw.B0, C0, nw)
    o.upload_cases(sw.w, sw.k, sw.depth, 1025.0, 9.81, sw.zeta, sw.beta)
    o.solve_dynamics_device(sw.nIter, sw.tol, sw.XiStart)      # warm-up (thread pool, page faults)
    t0 = time.perf_counter()
    o.solve_dynamics_device(sw.nIter, sw.tol, sw.XiStart)
    dt = time.perf_counter() - t0
    res = o.fetch_results(want_Xi=True)
    o.close()
    flops = algorithmic_flops(off, nw, res["niter"])
    base = {"value": n * nw / dt, "unit": "dcf solves/s", "cores": threads, "kind": "port",
            "sample": "all %d designs of the timed batch x 1 sea state x %d bins, oracle/raftx_oracle.c (gcc -O3 -march=x86-64-v3, "
                      "IEEE semantics, OpenMP over (design, case), %d threads), %.1f s" % (n, nw, threads, dt),
            "algorithmic_gflops": flops / dt / 1e9,
            "note": "a scalar loop-by-loop restatement of the reference (materialised kinematics, libm cabs): the checker doing "
                    "double duty, not a tuned CPU implementation -- %.2f GFLOP/s per thread" % (flops / dt / 1e9 / max(threads, 1))}
    return base, res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--designs", type=int, default=10000, help="designs per GPU (weak scaling)")
    ap.add_argument("--chunks", type=int, default=0, help="design blocks per step (0 = library default)")
    ap.add_argument("--workers", type=int, default=0, help="internal streams (0 = library default)")
    ap.add_argument("--xi-out", action="store_true", help="download the full responses inside the step as well")
    ap.add_argument("--pageable", action="store_true", help="descriptors in ordinary NumPy memory instead of page-locked")
    ap.add_argument("--resident", action="store_true", help="also time the fused kernel alone, whole batch resident in HBM")
    ap.add_argument("--no-stream", action="store_true", help="time isolated blocking calls (raftx_sweep_stats) instead of streaming the "
                                                             "steps through the library's two slots (raftx_sweep_submit / _wait)")
    ap.add_argument("--no-cpu-baseline", action="store_true", help="skip the oracle leg (cpu_baseline and the all-design check)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    # RAFTX_BENCH_BACKEND=gloo + RAFTX_BENCH_DEVICE=0: rehearsal of the multi-rank path on a single-GPU box (all ranks
    # share device 0, host transport for the gather); the driver's runs use RCCL with one GPU per rank
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
    from raft_amd.metrics import rao_group_err
    # the rank's host thread and page-locked buffers go onto the socket its GPU hangs off (raft_amd/locality.py);
    # the original CPU set comes back for the oracle leg, which wants every core
    from raft_amd import locality
    cpus_at_start = os.sched_getaffinity(0)
    placement = ({"bound": False, "why": "RAFTX_NO_BIND"} if os.environ.get("RAFTX_NO_BIND")
                 else locality.bind_near_device(backend.hip_library(), local))
    ctx = backend.hip_library().context(local)
    sw, fx, geo = make_sweep(ctx, args.designs, rank, pinned=not args.pageable)
    nw, nD = sw.nw, sw.n_design

    comm = None
    gather_kind = None
    ctx_comm = None
    if world > 1:
        from raft_amd import comm as rcomm
        # the exchange step gets its own context (= its own stream): the gather of step i must not queue behind the
        # kernels of step i + 1, which are already on the solver context's stream when the steps are streamed
        ctx_comm = backend.hip_library().context(local)
        comm, gather_kind = rcomm.from_env(ctx_comm, prefer="rccl" if backend_name == "nccl" else "host")

    stream_steps = not args.no_stream
    Xi_pinned = [ctx.pinned_empty((nD, 1, 1, 6, nw)) for _ in range(2 if stream_steps else 1)] if args.xi_out else [None, None]

    def gather(r):
        if comm is not None:                              # statistics of every rank onto rank 0 (48 B + 8 B per design-case)
            r["std_all"] = comm.gather_rows(np.concatenate([r["std"].reshape(nD, -1), r["niter"].reshape(nD, -1).astype(np.float64)], axis=1),
                                            counts=np.full(world, nD, dtype=np.int64))       # weak scaling: every rank holds nD designs
        return r

    def step():                                           # one isolated, blocking crossing
        return gather(sw.run_crossing(ctx, n_chunk=args.chunks, n_worker=args.workers, Xi_out=Xi_pinned[0]))

    def run_steps(n):
        """n whole solver-stage steps; returns the per-step results.  Streamed: step i + 1 is submitted (its descriptor
        upload and member pass start) before step i is collected, as consecutive batches of a long sweep are; every
        step still moves its own descriptors in and its own statistics out."""
        if not stream_steps:
            return [step() for _ in range(n)]
        out = []
        h = sw.submit_crossing(ctx, 0, n_chunk=args.chunks, Xi_out=Xi_pinned[0]) if n > 0 else None
        for i in range(n):
            h_next = sw.submit_crossing(ctx, (i + 1) % 2, n_chunk=args.chunks, Xi_out=Xi_pinned[(i + 1) % 2]) if i + 1 < n else None
            out.append(gather(sw.wait_crossing(ctx, h)))
            h = h_next
        return out

    def barrier():
        if dist is not None:
            import torch
            torch.cuda.synchronize()
            dist.barrier()
            torch.cuda.synchronize()

    if stream_steps:
        # untimed priming, before the W warm-up steps: the first crossing of a stream is cut into two blocks, the following
        # ones are single blocks, and each slot's block contexts allocate their device buffers the first time they meet a
        # configuration -- three streamed steps bring both slots to the steady-state one
        run_steps(3)
    run_steps(args.warmup)
    barrier()
    t0 = time.perf_counter()
    res = run_steps(args.steps)                           # returns after the last step's streams have drained
    barrier()
    elapsed = time.perf_counter() - t0
    r = res[-1]
    tims = [x["timing_ms"] for x in res]
    isolated = None
    if stream_steps and rank == 0:                        # the same step as an isolated blocking call (outside the timed region)
        t1 = time.perf_counter()
        for _ in range(5):                                # (no gather here: only this rank runs it)
            sw.run_crossing(ctx, n_chunk=args.chunks, n_worker=args.workers, Xi_out=Xi_pinned[0])
        isolated = (time.perf_counter() - t1) / 5
    if dist is not None:
        import torch
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if backend_name == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    tims = np.array(tims)
    off = r["strip_off"]
    niter = r["niter"]
    nan = int(np.count_nonzero(r["flags"] & 2))
    if comm is not None and rank == 0:
        assert r["std_all"].shape == (nD * world, 7) and np.array_equal(r["std_all"][:nD, :6], r["std"].reshape(nD, 6))

    # ---- parity of the timed batch (outside the timed region): one more crossing with the responses downloaded
    chk = sw.run_crossing(ctx, n_chunk=args.chunks, n_worker=args.workers, want_Xi=True)
    for x in res:                                         # every timed step produced the same bits
        assert np.array_equal(x["std"].view(np.uint64), r["std"].view(np.uint64)) and np.array_equal(x["niter"], r["niter"])
    assert np.array_equal(chk["std"].view(np.uint64), r["std"].view(np.uint64)) and np.array_equal(chk["niter"], niter), \
        "two crossings of the same batch differ"
    Xi = chk["Xi"]
    parity = {"nan_flags": nan}
    if rank == 0:                                         # the first 64 designs are the live reference's own variants
        errs, mism = [], 0
        for j, sol in enumerate(fx["solved"][:min(nD, 64)]):
            errs.append(rao_group_err(Xi[j, 0, 0], np.asarray(sol["Xi"])[0], sw.zeta[0, 0]))
            mism += int(int(niter[j, 0]) != int(sol["units"][0]["niter"]))
        parity["reference_solved_designs"] = len(errs)
        parity["rao_max_rel_err_vs_reference"] = float(max(errs)) if errs else None
        parity["niter_mismatches_vs_reference"] = mism
        assert mism == 0 and (not errs or max(errs) < 1e-6), "bench results fail parity against the reference: %r" % parity
    assert nan == 0, "NaN flags in the timed batch"

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        os.sched_setaffinity(0, cpus_at_start)
        cpu, ores = oracle_run(sw, ctx)
        e = [rao_group_err(Xi[d, 0, 0], ores["Xi"][d, 0, 0], sw.zeta[0, 0]) for d in range(nD)]
        parity["oracle_checked_designs"] = int(nD)
        parity["rao_max_rel_err_vs_oracle"] = float(np.max(e))
        parity["niter_mismatches_vs_oracle"] = int(np.count_nonzero(ores["niter"] != niter))
        assert parity["rao_max_rel_err_vs_oracle"] < 1e-6 and parity["niter_mismatches_vs_oracle"] == 0, \
            "bench results fail parity against the oracle: %r" % parity

    # ---- optional: the fused kernel alone on the whole batch, resident in / resident out
    resident = None
    if args.resident:
        sw.upload(ctx)
        ks = []
        for i in range(args.warmup + args.steps):
            ctx.solve_dynamics_device(sw.nIter, sw.tol, sw.XiStart)
            if i >= args.warmup:
                ks.append(ctx.last_kernel_ms())
        k = float(np.mean(ks))
        resident = {"kernel_ms": k, "dcf_per_s": nD * nw / (k * 1e-3),
                    "hbm_frac": algorithmic_bytes(off, nw) / (k * 1e-3) / 1e9 / HBM_PEAK_GBS,
                    "fp64_valu_frac": algorithmic_flops(off, nw, niter) / (k * 1e-3) / 1e12 / FP64_VALU_PEAK_TF}

    n_dcf_rank = nD * 1 * nw
    value = n_dcf_rank * world * args.steps / elapsed
    k_sum_ms = float(np.mean(tims[:, 2]))                 # k_solve_dynamics: summed HIP-event durations of one step's launches
    A = algorithmic_bytes(off, nw)
    flops = algorithmic_flops(off, nw, niter)
    out = {
        "metric": "design-case-frequency solves/sec (whole node)",
        "value": value, "unit": "dcf solves/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1e3 * elapsed / args.steps,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f64", "data": "synthetic",
        "config": {"workload": "C3 VolturnUS-S parameter sweep: %d designs/GPU x 1 sea state (JONSWAP Hs6 Tp12, 0 deg) x %d bins, "
                               "nIter=4, tol=0.01; designs = distinct U[0.75,1.25]^5 variants (default_rng(0))" % (nD, nw),
                   "designs_per_gpu": nD, "cases": 1, "nw": nw,
                   "step": "whole solver stage (SURVEY 8d): descriptor H2D + table/statics generation + fused fixed point + "
                           "statistics + D2H of %s; %s"
                           % ("statistics and full responses" if args.xi_out else "statistics (\"stats out\")",
                              "steps streamed through the library's two slots (raftx_sweep_submit / raftx_sweep_wait): the descriptor "
                              "upload and member pass of step i+1 run beside the kernels of step i, as consecutive batches of a long "
                              "sweep do; every step moves its own 66 MB in and its own statistics out" if stream_steps
                              else "isolated blocking calls (raftx_sweep_stats)"),
                   "streamed": bool(stream_steps),
                   "state": "xi out" if args.xi_out else "stats out",
                   "sharding": "designs over ranks, no collective while solving; statistics gathered to rank 0 inside the step"
                               if world > 1 else "single GPU",
                   "gather": gather_kind},
        "step_breakdown_ms": {("latency_submit_to_collected" if stream_steps else "wall_in_library"): float(np.mean(tims[:, 0])),
                              "generation_kernels_sum": float(np.mean(tims[:, 1])),
                              "solve_kernels_sum": k_sum_ms, "statistics_kernels_sum": float(np.mean(tims[:, 3]))},
        "parity": parity,
        "mean_iterations": float(np.mean(niter)),
        "roofline": {"bound": "hbm", "achieved": A / (k_sum_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": A / (k_sum_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": measured_traffic(nD),
                     "kernel": "k_solve_dynamics", "kernel_ms_per_step": k_sum_ms, "algorithmic_bytes_per_step": A,
                     "note": "summed over the launches of one step (the library cuts a step into design blocks); the fused "
                             "kernel is fp64-VALU-bound (SURVEY.md 8d): see roofline_fp64_valu"},
        "roofline_fp64_valu": {"achieved": flops / (k_sum_ms * 1e-3) / 1e12, "peak": FP64_VALU_PEAK_TF, "unit": "TFLOP/s",
                               "frac": flops / (k_sum_ms * 1e-3) / 1e12 / FP64_VALU_PEAK_TF,
                               "algorithmic_flops_per_step": flops,
                               "sustained_fma_probe": {"tflops": FP64_FMA_SUSTAINED_TF,
                                                       "frac": flops / (k_sum_ms * 1e-3) / 1e12 / FP64_FMA_SUSTAINED_TF,
                                                       "note": "what a pure v_fma_f64 loop sustains on an MI355X at the kernel's "
                                                               "occupancy (the clock drops to 1.7 GHz under fp64 load): "
                                                               "scripts/ubench/valu_mfma_probe.hip, profiles/r02_valu_mfma_probe.jsonl"}},
        "host_placement": placement,
        "geometry": dict(geo, strips=int(off[-1]), strip_table_bytes_not_uploaded=int(off[-1]) * 256),
    }
    if isolated is not None:
        out["isolated_call"] = {"ms_per_step": 1e3 * isolated, "dcf_per_s_per_gpu": n_dcf_rank / isolated,
                                "note": "the same step as one blocking raftx_sweep_stats call with nothing else in flight "
                                        "(upload of all descriptors on the critical path)"}
    if resident is not None:
        out["kernel_resident"] = resident
    ref_path = os.path.join(ROOT, "profiles", "reference_cpu_timing.json")
    if os.path.exists(ref_path):      # the unmodified NumPy reference, timed in the BUILD container (it cannot travel to the GPU box)
        with open(ref_path) as f:
            rt = json.load(f)
        out["reference_numpy_build_container"] = {"dcf_per_s_one_core": rt.get("dcf_per_s_per_core"),
                                                  "dcf_per_s_pool": rt.get("pool", {}).get("dcf_per_s_all_cores_solve_only"),
                                                  "pool_cores": rt.get("pool", {}).get("cores"),
                                                  "where": "BUILD CONTAINER (8 cores), second-hand here: the reference tree does not travel to the GPU box"}
    if cpu is not None:
        out["cpu_baseline"] = cpu
    if rank == 0:
        print(json.dumps(out))
    if comm is not None:
        comm.close()
    if ctx_comm is not None:
        ctx_comm.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
