"""Legs of bench.py outside the headline: BASELINE.json's configs[1], [3] and [4] at their
specified sizes, each checked against its committed live-reference golden and carrying the roofline of its own dominant
kernel.  Every function takes the rank's raftx context (and, for the sharded forms, a raft_amd.comm communicator) and
returns a JSON-ready dict; nothing here touches the oracle.  At the end: the legs of the C3 headline run itself (responses out,
host-made descriptors, launch size, the 1 250-design shard), which take bench.py's state as one namespace `B`.

Roofline accounting:
  k_solve_system_rows   one coupled solve of an n = 6 N system with R right-hand sides = n^3/3 complex multiply-adds of the
                        pivoted LU + n^2 R of the two triangular sweeps, 8 real FLOPs each: 41 472 FLOP at N = 4, R = 1
                        (36 864 of them the factorisation);
  k_qtf_pairs           2 464 fp64 FLOP per (strip, frequency pair): the executed count of the rocprofv3 counters on the
                        16-set VolturnUS-S launch (profiles/r02_c4_c5/qtf_pmc.json: 4.2e10 FLOP over 16 x 20 100 x 53),
                        used as the algorithmic figure of the slender-body pair expression (raft_member.py:1541-1668).
"""
import time

import numpy as np

FP64_VALU_PEAK_TF = 78.6
QTF_FLOP_PER_STRIP_PAIR = 2464.0


def system_solve_flops(n_unit, n_rhs=1):
    n = 6.0 * n_unit
    return 8.0 * (n ** 3 / 3.0 + n * n * n_rhs)


def _roof(flops, ms, kernel):
    tf = flops / (ms * 1e-3) / 1e12
    return {"bound": "fp64_valu", "kernel": kernel, "achieved": tf, "peak": FP64_VALU_PEAK_TF, "unit": "TFLOP/s",
            "frac": tf / FP64_VALU_PEAK_TF, "kernel_ms": ms, "algorithmic_flops": flops}


def c2_dropin(ctx, repeat=5):
    """configs[1]: examples/VolturnUS-S_example.yaml, 3 sea states x 200 bins through the drop-in Model.solveDynamics
    (raft/raft_model.py:994-1255) -- host packing of the strip table from the Member objects, upload, fused kernel, system
    solve, downloads: what a caller of the reference's method waits for -- against tests/golden/c2_volturnus.npz."""
    from raft_amd import dropin, snapshot
    from raft_amd.metrics import group_rel_err, rao_group_err
    fx, model = snapshot.load_model_fixture("c2_volturnus.npz")
    eng = dropin.Engine(ctx)
    rows = []
    for c in fx["cases"][:3]:
        case = snapshot.case_from_fixture(c)
        ts, ks = [], []
        Xi = None
        for i in range(repeat + 1):                       # the first call warms the context (allocations, code objects)
            t0 = time.perf_counter()
            Xi = eng.solveDynamics(model, dict(case)).copy()
            ts.append(time.perf_counter() - t0)
            ks.append(ctx.last_kernel_ms())
        nH = Xi.shape[0] - 1
        ref = np.asarray(c["Xi"])[:nH]
        u = c["units"][0]
        rows.append({"wave": [case.get("wave_height"), case.get("wave_period"), case.get("wave_heading")],
                     "call_ms_median": 1e3 * float(np.median(ts[1:])), "call_ms_min": 1e3 * float(np.min(ts[1:])),
                     "niter": int(model._raftx_niter[0]), "niter_reference": int(u["niter"]),
                     "max_group_rel_err_vs_reference": float(group_rel_err(Xi[:nH], ref)),
                     "rao_max_rel_err_vs_reference": float(rao_group_err(Xi[0], ref[0], np.asarray(u["zeta"])[0])),
                     "reference_numpy_s_build_container": float(c["ref_seconds"])})
        assert rows[-1]["niter"] == rows[-1]["niter_reference"] and rows[-1]["rao_max_rel_err_vs_reference"] < 1e-6, rows[-1]
    n_dcf = len(rows) * int(model.nw)
    total = sum(r["call_ms_median"] for r in rows) * 1e-3
    return {"config": "C2 VolturnUS-S_example.yaml: 3 sea states x %d bins, one drop-in Model.solveDynamics call each" % model.nw,
            "golden": "tests/golden/c2_volturnus.npz (live reference)", "cases": rows,
            "ms_per_call": 1e3 * total / len(rows), "dcf_per_s": n_dcf / total,
            "max_rao_rel_err_vs_reference": max(r["rao_max_rel_err_vs_reference"] for r in rows),
            "note": "one design, one sea state per call: 6 x 200 unknowns -- the call is host work (strip packing from the Member "
                    "objects, ctypes, 5 launches) around ~0.1 ms of kernels; the batched sweep (headline) is the throughput path"}


def _farm_fixture():
    from raft_amd import dropin, snapshot
    fx, model = snapshot.load_model_fixture("c4_farm.npz")   # nw = 200, the 50 seeded sea states of default_rng(1)
    cases = [snapshot.case_from_fixture(c) for c in fx["cases"]]
    assert len(cases) == 50 and model.nw == 200
    return fx, model, dropin.sweep_from_units(model, cases)


def c4_farm(ctx, farms=1000, repeat=3, comm=None):
    """configs[3]: the 4-unit VolturnUS-S farm (24-DOF block solve, raft/raft_model.py:1164-1236) x 200 bins x 50 sea
    states -- as specified (every sea state against the live reference, tests/golden/c4_farm.npz) and as a SWEEP of
    ``farms`` such farms (replicas of the layout: the coupled-solve kernel does not care that they are equal), which is
    what fills the chip: 4 F x 50 unit fixed points with resident Z / F_wave, then 50 F x 200 coupled solves in one launch.
    comm (world > 1): the specified farm's SEA STATES are block-partitioned over the ranks (SURVEY.md 8e C4,
    raft_amd.sweep.run_farm_sharded) and gathered on rank 0; the farm sweep is weak-scaled (``farms`` per rank)."""
    from raft_amd import snapshot
    from raft_amd.metrics import group_rel_err
    from raft_amd.sweep import Sweep, run_farm_sharded
    fx, model, sweep = _farm_fixture()
    nw = int(model.nw)
    Cc = fx["coupling_C"][None]
    world = 1 if comm is None else comm.world
    rank = 0 if comm is None else comm.rank
    walls = []
    out = None
    for _ in range(repeat + 1):
        t0 = time.perf_counter()
        out = run_farm_sharded(sweep, ctx, 4, Cc=Cc, comm=comm)
        walls.append(time.perf_counter() - t0)
    res = {"config": "C4 VolturnUS-S_farm.yaml: 4 units (24-DOF block solve) x %d bins x 50 sea states" % nw,
           "golden": "tests/golden/c4_farm.npz (live reference, all 50 sea states)", "n_gpus": world,
           "sharding": "sea states block-partitioned over ranks, responses gathered on rank 0" if world > 1 else "single GPU"}
    if rank == 0:
        err = max(group_rel_err(out["Xi"][0, i, :1], snapshot.ref_headings(c)[0]) for i, c in enumerate(fx["cases"]))
        nmis = sum(int(int(out["niter"][u, i]) != int(c["units"][u]["niter"])) for i, c in enumerate(fx["cases"]) for u in range(4))
        assert err < 1e-6 and nmis == 0, (err, nmis)
        res.update({"max_group_rel_err_vs_reference_all_50_sea_states": float(err), "niter_mismatches_vs_reference": nmis,
                    "as_specified": {"unit_fixed_points_kernel_ms": float(out["kernel_ms"][0]),
                                     "coupled_solves_kernel_ms": float(out["kernel_ms"][1]),
                                     "wall_ms_incl_upload_download_gather": 1e3 * float(np.median(walls[1:])),
                                     "dcf_per_s_wall": 4 * 50 * nw / float(np.median(walls[1:])),
                                     "roofline": _roof(50 * nw / world * system_solve_flops(4), float(out["kernel_ms"][1]), "k_solve_system_rows<4,1>")}})
    if farms > 0:
        rep = lambda a: None if a is None else np.concatenate([a] * farms, axis=0)
        off = np.concatenate([[0]] + [sweep.off[1:] + i * sweep.off[-1] for i in range(farms)])
        big = Sweep(off, rep(sweep.strips), rep(sweep.M0), rep(sweep.B0), rep(sweep.C0), sweep.w, sweep.k, sweep.depth, sweep.zeta,
                    sweep.beta, sweep.nIter, sweep.XiStart, tol=sweep.tol, MBw=rep(sweep.MBw))
        Ccb = np.repeat(Cc, farms, axis=0)
        ob = None
        ku, kc = [], []
        for i in range(repeat):
            ob = big.run_farm(ctx, 4, Cc=Ccb)
            if i:
                ku.append(ob["kernel_ms"][0])
                kc.append(ob["kernel_ms"][1])
        assert np.array_equal(ob["Xi"][0].view(np.uint64), ob["Xi"][farms - 1].view(np.uint64))
        if rank == 0 and world == 1:
            assert np.array_equal(ob["Xi"][0].view(np.uint64), out["Xi"][0].view(np.uint64)), "a farm of the sweep differs from the farm alone"
        n_solve = farms * 50 * nw
        k_u, k_c = float(np.mean(ku)), float(np.mean(kc))
        res["farm_sweep"] = {"farms_per_gpu": farms, "unit_pairs": 4 * farms * 50, "unit_fixed_points_kernel_ms": k_u,
                             "coupled_solves": n_solve, "coupled_solves_kernel_ms": k_c,
                             "coupled_solves_per_s": n_solve / (k_c * 1e-3),
                             "dcf_per_s_kernels": 4 * farms * 50 * nw / ((k_u + k_c) * 1e-3),
                             "roofline": _roof(n_solve * system_solve_flops(4), k_c, "k_solve_system_rows<4,1>"),
                             "every_farm_bit_identical_to_the_first": True}
    return res


def _qtf_sets(n_set, deck):
    """n_set (heading, motion) sets on the 200 x 200 second-order grid of SURVEY 8d C5 for ``deck``'s strip table."""
    from raft_amd import qtf as rq, waves, snapshot
    fx = snapshot.load_fixture(deck)
    f = snapshot.build_model(fx["model"]).fowtList[0]
    tab = rq.pack_qtf(f)
    nw2 = 200
    w2 = np.arange(1, nw2 + 1) * 0.0025 * 2 * np.pi
    k2 = np.array([waves.wave_number(x, f.depth) for x in w2])
    rng = np.random.default_rng(0)
    amp = np.array([1.0, 0.3, 0.7, 0.01, 0.02, 0.004])[:, None] / (1.0 + (w2[None, :] / 0.6) ** 2)
    Xi = np.array([amp * np.exp(1j * (rng.uniform(0, 6, 6)[:, None] + 1.5 * w2[None, :])) for _ in range(n_set)])
    betas = rng.uniform(0, 2 * np.pi, n_set)
    return f, tab, w2, k2, Xi, betas, np.array([f.M_struc] * n_set)


def c5_qtf_batch(ctx, n_set=16, repeat=4, deck="refgold_qtf_VolturnUS-S.npz", comm=None, kay=False):
    """One batch of slender-body QTFs on the 200 x 200 grid (20 100 upper-triangle pairs per set).  comm (world > 1): the
    ROWS of every matrix are interleaved over the ranks and the partial matrices SUM-reduced onto rank 0 (SURVEY.md 8e C5,
    raft_amd.sweep.run_qtf_rows_sharded).  Returns kernel / wall times of this rank and the result on rank 0."""
    from raft_amd.sweep import run_qtf_rows_sharded
    f, tab, w2, k2, Xi, betas, Ms = _qtf_sets(n_set, deck)
    tabs = [tab] * n_set
    ks, kk, walls = [], [], []
    q = None
    for i in range(repeat + 1):
        t0 = time.perf_counter()
        if kay:
            ctx.qtf_kay(tabs, betas, w2, k2, f.depth, f.rho_water, f.g)
            kk.append(ctx.last_kernel_ms())
        q = run_qtf_rows_sharded(ctx.qtf_slender, tabs, Xi, betas, w2, k2, f.depth, f.rho_water, f.g, Ms, None, comm=comm)
        walls.append(time.perf_counter() - t0)
        ks.append(ctx.last_kernel_ms())
    S = int(tab.strips.shape[0])
    pairs = n_set * 200 * 201 // 2
    return {"sets": n_set, "nw2": 200, "strips": S, "pairs": pairs, "qtf_kernels_ms": float(np.mean(ks[1:])),
            "kim_yue_kernels_ms": float(np.mean(kk[1:])) if kk else None, "wall_ms": 1e3 * float(np.median(walls[1:])), "q": q}


def c5_qtf(ctx, n_set=16):
    """configs[4]: examples/OC4semi-RAFT_QTF.yaml, 200 x 200 difference-frequency grid.  (i) parity and call time: the two
    sea states (0 and 30 deg) through the drop-in Model.solveDynamics with potSecOrder == 1 (first-order fixed point, Kim &
    Yue tables, QTF from the converged motions, second-order force, restarted fixed point -- raft_model.py:1108-1131,
    raft_fowt.py:1988-2078) against the live reference (tests/golden/c5_oc4semi_full.npz); (ii) the pair kernel's roofline
    on batches of ``n_set`` sets: the VolturnUS-S strip table the counters were taken on, and the OC4semi deck itself."""
    from raft_amd import dropin, snapshot
    from raft_amd.metrics import group_rel_err, rel_err
    fx, model = snapshot.load_model_fixture("c5_oc4semi_full.npz")
    f = model.fowtList[0]
    eng = dropin.Engine(ctx)
    iu = np.triu_indices(200)
    rows = []
    for c in fx["cases"]:
        case = snapshot.case_from_fixture(c)
        ts = []
        for _ in range(3):
            t0 = time.perf_counter()
            Xi = eng.solveDynamics(model, dict(case))
            ts.append(time.perf_counter() - t0)
        u = c["units"][0]
        nH = len(np.atleast_1d(u["beta"]))
        rows.append({"wave_heading": case.get("wave_heading"), "call_ms": 1e3 * float(np.min(ts[1:])),
                     "qtf_rel_err_vs_reference": float(rel_err(f.qtf[:, :, 0, :][iu], u["qtf_triu"])),
                     "Fhydro_2nd_rel_err_vs_reference": float(rel_err(f.Fhydro_2nd, u["Fhydro_2nd"])),
                     "response_group_rel_err_vs_reference": float(group_rel_err(Xi[:nH], np.asarray(c["Xi"])[:nH])),
                     "niter": int(model._raftx_niter[0]), "niter_reference": int(u["niter"]),
                     "reference_numpy_s_build_container": float(c.get("ref_seconds", float("nan")))})
        r = rows[-1]
        assert r["qtf_rel_err_vs_reference"] < 1e-6 and r["response_group_rel_err_vs_reference"] < 1e-6 and r["niter"] == r["niter_reference"], r
    v = c5_qtf_batch(ctx, n_set, deck="refgold_qtf_VolturnUS-S.npz")
    o = c5_qtf_batch(ctx, n_set, deck="c5_oc4semi_qtf.npz", kay=True)
    herm = bool(np.allclose(v["q"][0], np.conj(np.transpose(v["q"][0], (1, 0, 2))), atol=1e-6 * np.abs(v["q"][0]).max()))
    flops_v = v["pairs"] * v["strips"] * QTF_FLOP_PER_STRIP_PAIR
    flops_o = o["pairs"] * o["strips"] * QTF_FLOP_PER_STRIP_PAIR
    return {"config": "C5 OC4semi-RAFT_QTF.yaml: slender-body QTF on the 200 x 200 grid (20 100 pairs), sea state (6 m, 12 s) at 0 and 30 deg",
            "golden": "tests/golden/c5_oc4semi_full.npz (live reference, 7 min of reference time per case)",
            "dropin_potSecOrder1_calls": rows,
            "max_qtf_rel_err_vs_reference": max(r["qtf_rel_err_vs_reference"] for r in rows),
            "batch_volturnus": {"sets": v["sets"], "strips": v["strips"], "qtf_kernels_ms": v["qtf_kernels_ms"],
                                "strip_pairs_per_s": v["pairs"] * v["strips"] / (v["qtf_kernels_ms"] * 1e-3), "hermitian": herm,
                                "roofline": _roof(flops_v, v["qtf_kernels_ms"], "k_qtf_pairs (+ k_qtf_tables, same HIP-event bracket)")},
            "batch_oc4semi": {"sets": o["sets"], "strips": o["strips"], "qtf_kernels_ms": o["qtf_kernels_ms"],
                              "kim_yue_kernels_ms": o["kim_yue_kernels_ms"],
                              "one_200x200_qtf_ms": (o["qtf_kernels_ms"] + o["kim_yue_kernels_ms"]) / o["sets"],
                              "strip_pairs_per_s": o["pairs"] * o["strips"] / (o["qtf_kernels_ms"] * 1e-3),
                              "roofline": _roof(flops_o, o["qtf_kernels_ms"], "k_qtf_pairs (+ k_qtf_tables, same HIP-event bracket)")}}


def flex_sweep(ctx, n_unit=16):
    """The first widening beyond the rigid 6-DOF scope: the reference's flexible deck (tests/test_data/VolturnUS-S-flexible.yaml,
    150 reduced DOFs, 40 bins) -- ``n_unit`` units x 3 sea states as ONE batch (raft_amd/flex.py: the whole fixed point on the
    device, raftx_flex_solve -- node-by-node strip sweeps of the batch in one launch per iteration, the projections with the
    units' T as MFMA GEMM tiles, every impedance solve of an iteration in one launch) against the drop-in's one-case-at-a-time
    Model.solveDynamics and the live-reference golden."""
    from raft_amd import dropin, snapshot
    from raft_amd.metrics import rel_err
    fx, model = snapshot.load_model_fixture("flex_volturnus.npz")
    eng = dropin.Engine(ctx)
    base = snapshot.case_from_fixture(fx["cases"][0])
    cases = [base, dict(base, wave_height=4.0, wave_period=9.0, wave_heading=-20.0), dict(base, wave_height=1.0, wave_period=6.0)]
    single, nit = [], []
    t_single = 0.0
    for rep in range(2):
        single, nit = [], []
        t0 = time.perf_counter()
        for c in cases:
            single.append(eng.solveDynamics(model, dict(c)).copy())
            nit.append(int(model._raftx_niter[0]))
        t_single = (time.perf_counter() - t0) / len(cases)
    sw = dropin.flex_sweep_from_models([model] * n_unit, cases)
    out, ts = None, []
    for rep in range(6):                                    # (host-side calls of a few ms jitter on the GPU box: the median of five)
        t0 = time.perf_counter()
        out = sw.run(ctx)
        ts.append(time.perf_counter() - t0)
    t_batch = float(np.median(ts[1:]))
    err = max(rel_err(out["Xi"][d, ic, 0], single[ic][0]) for d in range(n_unit) for ic in range(3))
    Xr = np.asarray(fx["cases"][0]["Xi"])[:1]
    err_ref = float(rel_err(out["Xi"][0, 0, :1], Xr))
    assert err < 1e-9 and err_ref < 1e-7 and all(list(out["niter"][d]) == nit for d in range(n_unit)), (err, err_ref)
    pairs = n_unit * 3
    n = int(out["Xi"].shape[3])
    solves = (int(out["niter"].sum()) + pairs) * int(model.nw)          # one per (pair, iteration, bin) + the all-headings solve
    flops = solves * 8.0 * (n ** 3 / 3.0 + n * n)
    dev_ms = float(out["kernel_ms"][1])
    return {"config": "VolturnUS-S-flexible (150 reduced DOFs, %d bins): %d units x 3 sea states in one batch" % (model.nw, n_unit),
            "golden": "tests/golden/flex_volturnus.npz (live reference)", "dropin_ms_per_unit_case": 1e3 * t_single,
            "batch_ms": 1e3 * t_batch, "batch_ms_min_max": [1e3 * min(ts[1:]), 1e3 * max(ts[1:])], "batch_ms_per_unit_case": 1e3 * t_batch / pairs, "speedup_vs_dropin": t_single * pairs / t_batch,
            "kernel_ms_excitation_sweep": float(out["kernel_ms"][0]), "device_ms_fixed_point": dev_ms,
            "roofline": {"bound": "fp64_valu", "kernel": "k_solve_dense_reg2<5,10,16> (the span of the whole fixed point: + strip sweeps, projections, convergence tests)",
                         "achieved": flops / (dev_ms * 1e-3) / 1e12, "peak": FP64_VALU_PEAK_TF, "unit": "TFLOP/s", "frac": flops / (dev_ms * 1e-3) / 1e12 / FP64_VALU_PEAK_TF,
                         "algorithmic_flops": float(flops), "note": "n^3/3 + n^2 complex multiply-adds per solve (the pivoted LU's count; the kernel is a Gauss-Jordan sweep: n^3)"},
            "max_rel_err_vs_dropin": float(err), "rel_err_vs_reference": err_ref, "iterations": [int(x) for x in out["niter"][0]],
            "reference_numpy_s_per_case_build_container": float(fx["cases"][0].get("ref_seconds", float("nan")))}


# ---------------------------------------------------------------- legs of the C3 headline run (bench.py hands over its state `B`)
def xi_out(B):
    """SURVEY 8d's literal step on one GPU: the responses Xi downloaded inside the step (192 MB per 10 000 designs), four
    batches in flight through the staged calls, and the same step as isolated blocking calls."""
    import os
    import sys
    Xp = [B.ctx.pinned_empty((B.nD, 1, 1, 6, B.nw)) for _ in range(4)]

    def xi_steps(n):
        # four batches in flight, staged: batch i downloads (3.4-3.6 ms of PCIe: longer than a batch's kernels), batch
        # i+1 solves, batch i+2 is queued behind it, batch i+3 uploads its descriptors and runs its member pass.
        # prepare() never waits; launch(i+2) waits for a member pass that ran a step earlier; wait(i) for the download.
        # (With three in flight -- prepare(i+2) only after wait(i) -- the chain download -> upload -> member pass ->
        # fused kernel was serial: 5.0-5.1 ms per step, profiles/r04_xi_timeline.txt.)
        sub = lambda i: B.sw.prepare_crossing(B.ctx, i % 4, n_chunk=B.args.chunks, Xi_out=Xp[i % 4])
        hs = {i: sub(i) for i in range(min(n, 3))}
        for i in range(min(n, 2)):
            B.sw.launch_crossing(B.ctx, hs[i])
        for i in range(n):
            if i + 3 < n:
                hs[i + 3] = sub(i + 3)
            if i + 2 < n:
                B.sw.launch_crossing(B.ctx, hs[i + 2])
            B.sw.wait_crossing(B.ctx, hs.pop(i))
            if os.environ.get("RAFTX_BENCH_DEBUG"):
                print("  xi step %d collected at %.3f ms" % (i, 1e3 * time.perf_counter()), file=sys.stderr)
    xi_steps(9)                                       # untimed: every one of the four slots reaches its steady-state configuration
    # a streak of 60 batches (or K if larger): with four batches in flight the fill and the drain of the pipeline are
    # worth two steps (the first batch's upload and kernels, the last batch's download), which a long sweep does not see
    n_xi = max(B.args.steps, int(os.environ.get("RAFTX_BENCH_XI_STEPS", "60")))
    B.ctx.synchronize()
    t1 = time.perf_counter()
    xi_steps(n_xi)
    B.ctx.synchronize()
    t_xi = (time.perf_counter() - t1) / n_xi
    t1 = time.perf_counter()
    for _ in range(3):
        B.sw.run_crossing(B.ctx, n_chunk=B.args.chunks, n_worker=B.args.workers, Xi_out=Xp[0])
    t_xi_iso = (time.perf_counter() - t1) / 3
    for b_ in Xp:
        assert np.array_equal(b_.view(np.uint64), B.Xi.reshape(b_.shape).view(np.uint64)), "xi-out leg: responses differ from the checked batch"
    leg_ = {"state": "xi out: SURVEY 8d's literal step, H2D of the descriptors + kernels + D2H of Xi (%.0f MB per step, "
                       "page-locked destination)" % (Xp[0].nbytes / 1e6),
              "streamed_ms_per_step": 1e3 * t_xi, "streamed_dcf_per_s": B.nD * B.nw / t_xi, "streamed_steps": n_xi, "batches_in_flight": 4,
              "isolated_ms_per_step": 1e3 * t_xi_iso, "isolated_dcf_per_s": B.nD * B.nw / t_xi_iso}
    for b_ in Xp:
        B.ctx.free_pinned(b_)
    return leg_


def host_descriptors(B):
    """The rounds-1-4 form of the step on the same box: ONE batch expanded by NumPy outside the step, its 66 MB of descriptors
    uploaded by DMA in every step (no device-side expansion)."""
    sw_h, _, geo_h = B.make_sweep(B.ctx, B.args.designs, B.rank, pinned=True, rows=B.shard[B.rank], variants=False)

    def steps(n):
        out_ = []
        h_ = sw_h.submit_crossing(B.ctx, 0, n_chunk=B.args.chunks) if n > 0 else None
        for i in range(n):
            hn = sw_h.submit_crossing(B.ctx, (i + 1) % 2, n_chunk=B.args.chunks) if i + 1 < n else None
            out_.append(sw_h.wait_crossing(B.ctx, h_))
            h_ = hn
        return out_
    steps(3)
    steps(B.args.warmup)
    B.ctx.synchronize()
    t1 = time.perf_counter()
    rs = steps(B.args.steps)
    B.ctx.synchronize()
    dt_ = (time.perf_counter() - t1) / B.args.steps
    k_ = float(np.mean([x["timing_ms"][2] for x in rs]))
    fl_ = float(np.mean([B.algorithmic_flops(x["strip_off"], B.nw, x["niter"]) for x in rs]))
    assert np.array_equal(rs[-1]["std"].view(np.uint64), B.chk["std"].view(np.uint64)), "host-made and device-made descriptors give different statistics"
    for name in ("members", "stations", "caps", "member_off", "station_off", "cap_off"):
        a = getattr(sw_h.tables, name, None)
        if a is not None and a.size:
            try:
                B.ctx.free_pinned(a)
            except ValueError:
                pass
    return {"ms_per_step": 1e3 * dt_, "value": B.nD * B.nw / dt_, "kernel_ms_per_step": k_,
            "roofline_frac": fl_ / (k_ * 1e-3) / 1e12 / FP64_VALU_PEAK_TF, "host_descriptor_ms_per_batch": geo_h["host_descriptor_ms"],
            "descriptor_bytes_per_step": geo_h["descriptor_bytes"], "statistics_bit_identical_to_device_made_descriptors": True,
            "note": "the step of rounds 1-4 on this box: the STANDARD batch's descriptors, expanded once by NumPy outside the step, "
                    "uploaded by DMA in every step (same candidates every step); the headline's step writes them on the device "
                    "for NEW candidates every step -- ~0.09 ms of stores that land inside the running fused kernel"}


def launch_size(B):
    """The fused kernel against the size of its launch: the same designs' stream, 20 000 and 40 000 pairs resident."""
    res = {}
    for n_ in (20000, 40000):
        sw2, _, _ = B.make_sweep(B.ctx, n_, 0, pinned=False)
        sw2.upload(B.ctx)
        ks = []
        for i in range(4):
            B.ctx.solve_dynamics_device(sw2.nIter, sw2.tol, sw2.XiStart)
            if i:
                ks.append(B.ctx.last_kernel_ms())
        r2 = B.ctx.fetch_results(want_Xi=False)
        k2 = float(np.mean(ks))
        fl2 = B.algorithmic_flops(sw2.off, B.nw, r2["niter"])
        res[str(n_)] = {"pairs_per_launch": n_, "kernel_ms": k2, "us_per_pair": 1e3 * k2 / n_, "mean_iterations": float(np.mean(r2["niter"])),
                        "dcf_per_s": n_ * B.nw / (k2 * 1e-3), "fp64_valu_frac": fl2 / (k2 * 1e-3) / 1e12 / FP64_VALU_PEAK_TF}
        del sw2, r2
    res["note"] = ("the SAME kernel on launches of 20 000 / 40 000 pairs (resident in, resident out): the drain of the last residency round and the "
                   "idle of a slot between workgroups weigh less, the clock is a few per cent higher -- T(n) ~ 0.33 ms + 0.251 us n (profiles/r05_launch_size_scaling.json)")
    return res


def shard_1250(B, ms_whole_sweep):
    """One rank's share of BASELINE configs[2] cut into 8 shards (raft/parametersweep.py:39-100): the streamed step at
    nD / 8 designs on ONE GPU; ms_whole_sweep = this run's step of the whole sweep, in ms."""
    n_sh = max(1, B.nD // 8)
    sw_s, _, _ = B.make_sweep(B.ctx, n_sh, 0, pinned=not B.args.pageable, rows=(0, n_sh), variants=B.variants)
    d = B.args.depth
    bno = {"next": 1}

    def fresh_s():
        if B.variants:
            b = bno["next"]
            bno["next"] += 1
            sw_s.set_params(B.G_.volturnus_params(B.scale_rows(b * n_sh, (b + 1) * n_sh)))

    def steps_s(n):
        return B.run_streamed(sw_s, B.ctx, n, d, fresh=fresh_s)
    steps_s(max(12, 2 * d + 1))
    n_t = max(B.args.steps, 60)
    B.ctx.synchronize()
    t1 = time.perf_counter()
    rs = steps_s(n_t)
    B.ctx.synchronize()
    dt_ = (time.perf_counter() - t1) / n_t
    k_ = float(np.mean([x["timing_ms"][2] for x in rs]))
    fl_ = float(np.mean([B.algorithmic_flops(x["strip_off"], B.nw, x["niter"]) for x in rs]))
    t10 = ms_whole_sweep * 1e-3
    return {"designs_per_step": n_sh, "ms_per_step": 1e3 * dt_, "value": n_sh * B.nw / dt_, "steps": n_t, "kernel_ms_per_launch": k_,
            "roofline_frac": fl_ / (k_ * 1e-3) / 1e12 / FP64_VALU_PEAK_TF,
            "step_frac": fl_ / dt_ / 1e12 / FP64_VALU_PEAK_TF,
            "ms_per_step_of_the_whole_sweep_on_this_gpu": 1e3 * t10,
            "projected_8_gpu_strong_speedup": t10 / dt_,
            "ideal_ms_per_step": 1e3 * t10 / 8,
            "note": "one rank's share of BASELINE configs[2] cut into 8 shards, streamed like the headline step.  1 250 pairs on "
                    "1 024 resident workgroup places are two residency rounds whatever the launch form (the persistent grid claims "
                    "them, the second round runs at a quarter of the chip's occupancy): 0.41 us per pair against 0.26 in a long "
                    "launch -- DESIGN.md 7, profiles/r06_experiments/"}
