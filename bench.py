#!/usr/bin/env python
"""bench.py -- design-case-frequency (dcf) solves / second on MI355X.

Workload (BASELINE.json configs[2], the one the north-star target is quoted
on; SURVEY.md 8d "C3"): a VolturnUS-S geometry sweep, nDesign DISTINCT variants
of the five parameters of raft/parametersweep.py:33-37, each x U[0.75,1.25]
(default_rng(0)), x 1 sea state (JONSWAP Hs 6 m, Tp 12 s, head seas) x 200
frequency bins, nIter=4 (up to 5 fixed-point iterations), all fp64.

A STEP is one whole pass of the solver stage as SURVEY.md 8d defines it --
"H2D of the tables + kernels + D2H" -- over a batch of NEW candidates: the five
sweep parameters of every design of this rank go in (raft/parametersweep.py:39-40
draws exactly these), the library writes the member descriptions in HBM
(raftx_sweep_prepare_variants -> k_geom_expand: the dependent-geometry edits of
parametersweep.py:56-87), generates strip tables / statics (k_geom_*), runs the
fused fixed point (k_solve_dynamics: strip sweeps + drag-linearisation iterations +
per-bin 6x6 complex solves) and the statistics kernel, and the response statistics
(std of the six motions + iteration counts + flags) come out; block-pipelined over
internal streams, consecutive steps streamed through the library's slots.  Every
timed step solves DIFFERENT candidates (rows of one default_rng(0) stream); the host
work per step is drawing the parameter rows (geometry.host_params_ms_per_step, inside
the timed loop).  `value` = dcf of all ranks / wall time of the K timed steps.
--descriptors host is the form of rounds 1-4: ONE batch expanded by NumPy
(geometry.host_descriptor_ms, outside the step) and its 66 MB re-uploaded every step.  ("state": "stats
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
  cpu_baseline      the UNMODIFIED reference (raft.Model.solveDynamics, NumPy/SciPy; kind "reference") timed on this
                    host's cores -- one core, and a multiprocessing.Pool over as many CPUs as the container can run on --
                    from the byte-compiled archive oracle/_ref/raft_reference.zip (oracle/stage_reference.py); beside it
                    (`port_simd`) the vectorised C port of the oracle with OpenMP.

  xi_out / isolated_call   the same step with the 192 MB of responses downloaded too
                    (SURVEY.md 8d's literal "D2H of Xi"), streamed and as isolated blocking
                    calls -- extra legs after the timed region, N = 1 only;
  featured_sweeps   the lean featured specialisations of the fused kernel on the same
                    designs (frequency-dependent M/B, three sea states, two headings,
                    MacCamy-Fuchs columns): resident kernel time per pair-iteration
                    against the plain sweep's, a sample checked against the oracle.

  launch_size       the same fused kernel on resident launches of 20 000 and 40 000 pairs: what a launch costs beyond its pairs
                    (its last residency round drains, and a long launch runs at a few per cent more clock);
  roofline.traffic  the fused kernel's L2 <-> fabric bytes per launch, measured BY this run (N = 1): two child runs of this script
                    under `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` (separate passes, counters only; live_traffic);
                    the committed profile's figure, labelled as such, where that is not possible.

  c2_dropin / c4_farm / c5_qtf   BASELINE.json's configs[1], [3], [4] at their specified sizes (bench_legs.py),
                    each against its committed live-reference golden and with the roofline of its own kernel;
  flex_sweep        a batch of units with flexible members (150 reduced DOFs) against the drop-in and its golden.

Multi-GPU: one process per GPU; designs are block-partitioned over ranks, no
collective while solving; the statistics are gathered onto rank 0 INSIDE the
timed region.  Barrier, max-over-ranks and gather all go through the library's
own communicator (raft_amd/comm.py: RCCL via raftx_comm_*, rendezvous over TCP on
MASTER_ADDR) -- no torch in this file.  Weak scaling: nDesign per rank is fixed.
`python bench.py --gpus N` with no WORLD_SIZE in the environment starts the N
ranks itself (launch_ranks: RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR /
MASTER_PORT + a random RAFTX_COMM_TOKEN per job); under an external launcher
(`python -m torch.distributed.run ... bench.py --gpus N`) the ranks are already
there and --gpus must equal WORLD_SIZE.  `--workload c4 | c5` run the sharded
forms of configs[3] (sea states over ranks, gather) and configs[4] (QTF rows
interleaved over ranks, one SUM-reduce) instead of the C3 sweep.

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


def scale_rows(lo, hi):
    """Rows [lo, hi) of THE default_rng(0) stream of U[0.75,1.25]^5 scale factors, without drawing the rows
    before them: PCG64 advanced by one step per double, the same numbers as default_rng(0).uniform(size=(hi, 5))[lo:]."""
    bg = np.random.PCG64(0)
    bg.advance(int(lo) * 5)
    return np.random.Generator(bg).uniform(0.75, 1.25, size=(int(hi) - int(lo), 5))


def make_sweep(ctx, n_design, rank=0, pinned=True, mcf=False, zeta=None, beta=None, MBw=None, rows=None, variants=False):
    """Descriptors of this rank's designs (host, vectorised): rank r takes rows [r*n, (r+1)*n) of one default_rng(0)
    draw (weak scaling), or the rows [lo, hi) given as ``rows`` (strong scaling: contiguous shards of ONE sweep), so rank
    0's first 64 designs are the committed reference-built variants.  mcf / zeta, beta / MBw: the featured
    legs (MacCamy-Fuchs columns; other sea states / headings; frequency-dependent added mass and damping)."""
    from raft_amd import snapshot
    from raft_amd import geometry as G
    from raft_amd.sweep import GeometrySweep
    fx = snapshot.load_fixture("c3_variants.npz")
    fg = snapshot.load_fixture("geom_units.npz")
    base = json.loads(fg["c3_base_json"])
    if mcf:                                             # MacCamy-Fuchs correction on the four columns (raft_member.py:1415-1420)
        for m in base["platform"]["members"][:2]:
            m["MCF"] = True
    u0 = [u for u in fg["units"] if u["name"] == "C3-variant-0"][0]
    # constants that are not geometry: rotor-nacelle assembly (live reference minus its massless-RNA twin), mooring
    M_rna = np.asarray(u0["M_struc"]) - np.asarray(u0["M_struc_bare"])
    C_rest = np.asarray(u0["C_struc"]) - np.asarray(u0["C_struc_bare"]) + np.diag([7e4, 7e4, 0.0, 0.0, 0.0, 1e8])
    lo, hi = (rank * n_design, (rank + 1) * n_design) if rows is None else (int(rows[0]), int(rows[1]))
    n_design = hi - lo
    scales = scale_rows(lo, hi)
    prog = None
    if variants:
        # the candidates as PARAMETERS: the library writes their descriptors on the device (raftx_sweep_prepare_variants);
        # what the host does per batch is draw / receive the parameter rows
        prog = G.volturnus_program(base)
        t0 = time.perf_counter()
        params = G.volturnus_params(scales)
        t_desc = time.perf_counter() - t0
        D = None
    else:
        t0 = time.perf_counter()
        D = G.volturnus_sweep(base, scales).tables()
        t_desc = time.perf_counter() - t0
    if pinned and D is not None:                        # page-locked staging (raftx_host_alloc): full-rate, asynchronous H2D
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
    zeta_ = np.asarray(fx["zeta"])[None] if zeta is None else zeta
    beta_ = np.asarray(fx["beta"])[None] if beta is None else beta
    if variants:
        from raft_amd.sweep import VariantSweep
        assert MBw is None
        sw = VariantSweep(prog, params, M0, B0, C0, fx["w"], fx["k"], float(fx["depth"]), zeta_, beta_, int(fx["nIter"]),
                          float(fx["XiStart"]), tol=0.01, add_mask=G.ADD_MORISON | G.ADD_HYDROSTATIC | G.ADD_INERTIA)
        b = prog.base
        geo = {"designs": int(n_design), "members": int(b.n) * int(n_design), "host_descriptor_ms": 1e3 * t_desc,
               "descriptors": "written on the device (k_geom_expand) from %d parameters per design; the host sends %d B per design "
                              "instead of %d" % (prog.n_param, 8 * prog.n_param, b.members.nbytes + b.stations.nbytes + b.caps.nbytes),
               "descriptor_bytes": int(params.nbytes),
               "descriptor_bytes_written_on_device": int(n_design) * int(b.members.nbytes + b.stations.nbytes + b.caps.nbytes),
               "descriptors_page_locked": False}
        return sw, fx, geo
    sw = GeometrySweep(D, M0, B0, C0, fx["w"], fx["k"], float(fx["depth"]), zeta_, beta_,
                       int(fx["nIter"]), float(fx["XiStart"]), tol=0.01, add_mask=G.ADD_MORISON | G.ADD_HYDROSTATIC | G.ADD_INERTIA,
                       MBw=MBw)
    geo = {"designs": int(n_design), "members": int(D.member_off[-1]), "host_descriptor_ms": 1e3 * t_desc,
           "descriptors": "expanded on the host (NumPy) and uploaded", "descriptor_bytes": int(D.members.nbytes + D.stations.nbytes + D.caps.nbytes),
           "descriptors_page_locked": bool(pinned)}
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


def live_traffic(n_design, timeout_s=120):
    """`roofline.traffic` measured IN this run: two child runs of this script (five whole-batch steps, `--profile`) under
    `rocprofv3 --pmc FETCH_SIZE` and `--pmc WRITE_SIZE` -- separate passes, no trace domains beside the counters, run from
    /tmp with TMPDIR=/tmp -- and the per-launch means of the fused kernel's whole-batch launches, FETCH_SIZE doubled as
    MI355X_MICROARCH.md prescribes for gfx950 (rocprofv3 reports KB).  Returns (bytes per launch, provenance) or
    (None, why): no rocprofv3, a pass that fails or takes longer than timeout_s, or this process itself running under a
    profiler.  The caller then falls back to the committed profile's figure and says so."""
    import csv, glob, shutil, signal, subprocess, tempfile
    if any(k.startswith(("ROCPROF", "ROCP_TOOL", "ROCPROFILER")) for k in os.environ) or "rocprof" in os.environ.get("LD_PRELOAD", ""):
        return None, "this process runs under a profiler"
    exe = shutil.which("rocprofv3") or ("/opt/rocm/bin/rocprofv3" if os.path.exists("/opt/rocm/bin/rocprofv3") else None)
    if not exe:
        return None, "rocprofv3 not found"
    tmp = tempfile.mkdtemp(prefix="raftx_pmc_", dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp")
    got, t_all = {}, time.perf_counter()
    try:
        for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
            d = os.path.join(tmp, ctr)
            cmd = [exe, "--pmc", ctr, "--output-format", "csv", "-d", d, "-o", "bench", "--", sys.executable, os.path.join(ROOT, "bench.py"),
                   "--steps", "5", "--warmup", "1", "--profile", "--designs", str(n_design)]
            pr = subprocess.Popen(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, start_new_session=True)
            try:
                rc = pr.wait(timeout=timeout_s)
            except subprocess.TimeoutExpired:
                os.killpg(pr.pid, signal.SIGKILL)             # the session this call started, nothing else
                pr.wait()
                return None, "the %s pass took longer than %d s" % (ctr, timeout_s)
            if rc != 0:
                return None, "the %s pass ended with rc=%d" % (ctr, rc)
            rows = []
            for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
                with open(f) as fh:
                    rows += [r for r in csv.DictReader(fh) if ("k_solve_dynamics" in r["Kernel_Name"] or "raftx_kp_f" in r["Kernel_Name"]) and r["Counter_Name"] == ctr]
            gmax = max((int(r["Grid_Size"]) for r in rows), default=0)
            vals = [float(r["Counter_Value"]) for r in rows if int(r["Grid_Size"]) == gmax]
            if not vals:
                return None, "the %s pass recorded no launch of the fused kernel" % ctr
            got[ctr] = (1e3 * float(np.mean(vals)), len(vals), gmax)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    f, w_ = got["FETCH_SIZE"][0], got["WRITE_SIZE"][0]
    return 2.0 * f + w_, {"measured_in_this_run": True, "how": "two child runs of bench.py --steps 5 --warmup 1 --profile under rocprofv3 --pmc FETCH_SIZE / "
                          "--pmc WRITE_SIZE (separate passes, counters only), means over the whole-batch launches of k_solve_dynamics; "
                          "traffic = 2 x FETCH_SIZE + WRITE_SIZE (FETCH_SIZE doubled per MI355X_MICROARCH.md for gfx950; Infinity-Cache hits count as traffic)",
                          "FETCH_SIZE_bytes_raw": f, "WRITE_SIZE_bytes_raw": w_, "launches": [got["FETCH_SIZE"][1], got["WRITE_SIZE"][1]],
                          "grid_size": got["FETCH_SIZE"][2], "designs_per_launch": int(n_design), "seconds": time.perf_counter() - t_all}


def traffic_provenance():
    """Provenance of the FALLBACK figure of `roofline.traffic` (no rocprofv3 on the host, N > 1, a child pass that failed):
    the committed profile's, scaled to the designs of one step.  The default N = 1 run measures it itself (live_traffic)."""
    path = os.path.join(ROOT, "profiles", "traffic_latest.json")
    if not os.path.exists(path):
        return None
    with open(path) as f:
        t = json.load(f)
    return {"measured_in_this_run": False, "file": "profiles/traffic_latest.json", "source": t.get("source"),
            "measured_at": t.get("measured_at"), "kernel_source_head": t.get("kernel_source_head"),
            "designs_per_launch_in_profile": t.get("designs_per_launch", t.get("designs_per_gpu"))}


def algorithmic_bytes(off, nw):
    """A_min = 256*S + 3*288 + 8*nw + 96*nw per (design, case)."""
    S = np.diff(off).astype(np.float64)
    return float(np.sum(256.0 * S + 864.0 + 104.0 * nw))


def algorithmic_flops(off, nw, niter):
    """N_it*(175*S + 2000) + 160*S fp64 flops per dcf (transcendentals not counted)."""
    S = np.diff(off).astype(np.float64)
    return float(np.sum(niter.reshape(-1) * (175.0 * S + 2000.0) + 160.0 * S) * nw)


def oracle_run(sw, ctx):
    """The CPU oracle (oracle/raftx_oracle.c) on this host, over EVERY design of the timed batch.  Two checks and one
    timing: (i) the oracle's own geometry chain (raftx_build_designs of oracle/raftx_geom_oracle.h) on the same member
    descriptions, compared with the strip tables and statics the device generated -- all designs; (ii) the oracle's
    fixed point on those tables: its responses check the batch, its wall time is the cpu_baseline."""
    import subprocess
    from raft_amd._abi import RaftxLib
    so = os.path.join(ROOT, "oracle", "libraftx_oracle_fast.so")          # same source as the checker, -O3 -march=x86-64-v3
    if not os.path.exists(so):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")], stdout=subprocess.DEVNULL)
    lib = RaftxLib(so)
    lib.lib.raftx_oracle_threads.restype = int
    import ctypes as C
    budget = cpu_budget()
    try:                                                  # one OpenMP thread per CPU the container can actually run on
        nthr = max(1, int(round(budget["effective_parallel_cpus"])))
        if budget.get("cgroup_cpu_limit"):
            nthr = max(1, min(nthr, int(budget["cgroup_cpu_limit"])))
        C.CDLL("libgomp.so.1").omp_set_num_threads(nthr)
    except OSError:
        pass
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
    # (i) generator check: the same descriptors through the oracle's serial member / strip walk
    t0 = time.perf_counter()
    sw.upload(o)
    t_geom = time.perf_counter() - t0
    o_strips, _ = o.fetch_strips(sw.off[-1])
    oS = o.fetch_statics()
    gen = {"designs": int(n), "strip_count_mismatches": int(np.count_nonzero(np.asarray(sw.off) != np.asarray(off))),
           "oracle_geometry_s": t_geom}
    if gen["strip_count_mismatches"] == 0:
        # the 26 geometric / coefficient fields group-wise relative to the group's largest magnitude, the member / strip indices
        # exactly (the run hints behind them are never read by the library: it detects runs itself)
        groups = [(0, 3), (3, 6), (6, 15), (15, 18), (18, 19), (19, 23), (23, 26)]      # positions, arms, triads, scalars (tests/test_geometry.py)
        gen["strips_max_err_rel_to_field_max"] = float(max(
            np.max(np.abs(strips[:, a:b] - o_strips[:, a:b])) / max(np.max(np.abs(o_strips[:, a:b])), 1e-300) for a, b in groups))
        gen["strip_index_mismatches"] = int(np.count_nonzero(strips[:, 26:28] != o_strips[:, 26:28]))
        for key in ("A_morison", "C_hydro", "M_struc", "C_struc", "W_hydro", "W_struc"):
            a, b = np.asarray(S[key]).reshape(n, -1), np.asarray(oS[key]).reshape(n, -1)
            gen["%s_max_group_rel_err" % key] = float(np.max(np.max(np.abs(a - b), axis=1) / np.maximum(np.max(np.abs(b), axis=1), 1e-300)))
    sw.off = off
    # (ii) the fixed point on the device-generated tables
    o.upload_designs_raw(off, strips, M0, sw.B0, C0, nw)
    o.upload_cases(sw.w, sw.k, sw.depth, 1025.0, 9.81, sw.zeta, sw.beta)
    o.solve_dynamics_device(sw.nIter, sw.tol, sw.XiStart)      # warm-up (thread pool, page faults)
    t0 = time.perf_counter()
    o.solve_dynamics_device(sw.nIter, sw.tol, sw.XiStart)
    dt = time.perf_counter() - t0
    res = o.fetch_results(want_Xi=True)
    flops = algorithmic_flops(off, nw, res["niter"])
    plain = {"value": n * nw / dt, "unit": "dcf solves/s", "cores": threads, "kind": "port",
             "sample": "all %d designs of the timed batch x 1 sea state x %d bins, oracle/raftx_oracle.c solve_core (gcc -O3 -march=x86-64-v3, "
                       "IEEE semantics, OpenMP over (design, case), %d threads), %.1f s" % (n, nw, threads, dt),
             "algorithmic_gflops": flops / dt / 1e9, "gflops_per_thread": flops / dt / 1e9 / max(threads, 1),
             "note": "the CHECKER timed: a scalar loop-by-loop restatement of the reference (materialised complex kinematics, libm "
                     "cabs, one malloc'ed work set per pair) -- not a tuned CPU implementation"}
    # the honest CPU datapoint: the same fixed point restated for the vector units (oracle/raftx_port_simd.h), checked against
    # the plain oracle's results of this very batch before its time counts
    import ctypes as C
    fn = lib.lib.raftx_oracle_solve_simd
    fn.argtypes = [C.c_void_p, C.c_int, C.c_double, C.c_double]
    fn.restype = C.c_int
    assert fn(o._h, int(sw.nIter), float(sw.tol), float(sw.XiStart)) == 0          # warm-up
    t0 = time.perf_counter()
    assert fn(o._h, int(sw.nIter), float(sw.tol), float(sw.XiStart)) == 0
    dts = time.perf_counter() - t0
    rs = o.fetch_results(want_Xi=True)
    # ... and on ONE thread (the first 128 designs): the per-core figure, independent of how many of the host's hardware
    # threads this container is actually allowed to run on
    one = None
    try:
        gomp = C.CDLL("libgomp.so.1")
        n1 = min(n, 128)
        o.upload_designs_raw(off[:n1 + 1], strips[:off[n1]], M0[:n1], sw.B0[:n1], C0[:n1], nw)
        gomp.omp_set_num_threads(1)
        fn(o._h, int(sw.nIter), float(sw.tol), float(sw.XiStart))
        t0 = time.perf_counter()
        fn(o._h, int(sw.nIter), float(sw.tol), float(sw.XiStart))
        dt1 = time.perf_counter() - t0
        gomp.omp_set_num_threads(threads)
        f1 = algorithmic_flops(off[:n1 + 1], nw, res["niter"][:n1])
        one = {"designs": int(n1), "dcf_per_s": n1 * nw / dt1, "algorithmic_gflops": f1 / dt1 / 1e9, "seconds": dt1}
    except Exception as e:                                   # noqa: BLE001 -- a missing libgomp only loses this datapoint
        one = {"error": "%s: %s" % (type(e).__name__, str(e)[:120])}
    o.close()
    num = np.max(np.abs(rs["Xi"] - res["Xi"]).reshape(n, -1), axis=1)
    den = np.max(np.abs(res["Xi"]).reshape(n, -1), axis=1)
    simd_err = float(np.max(num / den))
    simd_mis = int(np.count_nonzero(rs["niter"] != res["niter"]))
    assert simd_err < 1e-10 and simd_mis == 0, "the vectorised CPU port differs from the oracle: %g, %d" % (simd_err, simd_mis)
    base = {"value": n * nw / dts, "unit": "dcf solves/s", "cores": threads, "kind": "port-simd",
            "sample": "all %d designs of the timed batch x 1 sea state x %d bins, oracle/raftx_port_simd.h (the oracle's fixed point with "
                      "frequency as the unit-stride inner loop, split re / im arrays, #pragma omp simd, AVX2 + FMA; OpenMP over (design, case), "
                      "%d threads), %.2f s" % (n, nw, threads, dts),
            "algorithmic_gflops": flops / dts / 1e9, "gflops_per_thread": flops / dts / 1e9 / max(threads, 1),
            "max_rel_err_vs_plain_oracle": simd_err, "niter_mismatches_vs_plain_oracle": simd_mis,
            "one_thread": one, "plain_oracle": plain}
    return base, res, gen


_CPU_BUDGET = {}


def cpu_budget():
    """oracle/time_reference.py --probe in a child process (never a fork of this process: it holds a HIP context): logical
    CPUs, the CPUs' worth of run time the container actually gets, the cgroup quota if one is visible."""
    import subprocess
    if not _CPU_BUDGET:
        try:
            r = subprocess.run([sys.executable, os.path.join(ROOT, "oracle", "time_reference.py"), "--probe"], timeout=120,
                               capture_output=True, text=True, check=True).stdout.strip().splitlines()[-1]
            _CPU_BUDGET.update(json.loads(r))
        except Exception as e:                              # noqa: BLE001
            n = len(os.sched_getaffinity(0))
            _CPU_BUDGET.update({"logical_cpus": n, "effective_parallel_cpus": float(n), "cgroup_cpu_limit": None,
                                "error": "%s: %s" % (type(e).__name__, str(e)[:120])})
    return dict(_CPU_BUDGET)


def reference_on_this_host():
    """The UNMODIFIED NumPy/SciPy reference (raft.Model.solveDynamics, raft/raft_model.py:966) timed on THIS host by
    oracle/time_reference.py in child processes: (i) one process, one core, 2 designs; (ii) multiprocessing.Pool over
    every usable core with 2 x cores (design, case) items, BLAS/OpenMP threads = 1.  The
    reference comes from /root/reference (build container) or from oracle/_ref/raft_reference.zip (GPU box: the
    byte-compiled archive oracle/stage_reference.py builds, git-ignored, shipped with the snapshot).  Each timed solve is
    compared with the committed live-reference fixture tests/golden/c3_variants.npz inside the child; a mismatch fails
    the leg.  Returns None when neither form of the reference is on this host."""
    import subprocess
    sys.path.insert(0, ROOT)
    from oracle import ref_harness as rh
    if not rh.reference_available():
        return None
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    script = os.path.join(ROOT, "oracle", "time_reference.py")
    env = dict(os.environ, OMP_NUM_THREADS="1", OPENBLAS_NUM_THREADS="1", MKL_NUM_THREADS="1", MPLBACKEND="Agg")
    out = {}
    try:
        t0 = time.perf_counter()
        # how many CPUs this container can RUN on at once (the pool's boxes show 256 logical CPUs to a container with a much
        # smaller CPU-time quota: Pool(256) measured there = 256 processes taking turns, 65 s per solve instead of 2.1)
        probe = cpu_budget()
        procs = max(1, min(cores, int(round(probe["effective_parallel_cpus"]))))
        if probe.get("cgroup_cpu_limit"):                  # the CFS quota is the sustained figure (the one-second probe sees its burst)
            procs = max(1, min(procs, int(probe["cgroup_cpu_limit"])))
        try:                                                # ~0.4 GB per worker (numpy + scipy + matplotlib + a Model)
            with open("/proc/meminfo") as f:
                avail_kb = [int(l.split()[1]) for l in f if l.startswith("MemAvailable")][0]
            procs = max(1, min(procs, int(avail_kb / 1024 / 400)))
        except Exception:                                   # noqa: BLE001
            pass
        r1 = json.loads(subprocess.run([sys.executable, script, "--designs", "2"], env=env, timeout=300, capture_output=True,
                                       text=True, check=True).stdout.strip().splitlines()[-1])
        rp = json.loads(subprocess.run([sys.executable, script, "--pool", "--procs", str(procs), "--items", str(2 * procs)],
                                       env=env, timeout=900, capture_output=True, text=True, check=True).stdout.strip().splitlines()[-1])
        err = max(r1["max_rel_err_vs_committed_reference_fixture"], rp["max_rel_err_vs_committed_reference_fixture"])
        assert err < 1e-9, "the timed reference does not reproduce tests/golden/c3_variants.npz: %g" % err
        out = {"dcf_per_s_one_core": r1["dcf_per_s_one_core"], "solveDynamics_s_per_design": r1["solveDynamics_s_per_design"],
               "model_build_s_per_design": r1["model_build_s_per_design"], "one_core_designs": r1["designs"],
               "dcf_per_s_pool": rp["dcf_per_s_pool"], "dcf_per_s_pool_solve_only": rp["dcf_per_s_pool_solve_only"],
               "pool_procs": rp["procs"], "pool_items": rp["items"], "pool_wall_s": rp["wall_s"],
               "pool_mean_solveDynamics_s": rp["mean_solveDynamics_s"], "pool_worker_busy_fraction": rp["worker_busy_fraction"],
               "cores_on_host": cores, "cpu_budget_probe": probe, "reference_from": rp["reference_from"],
               "max_rel_err_vs_committed_reference_fixture": err, "leg_wall_s": time.perf_counter() - t0}
        return out
    except subprocess.CalledProcessError as e:
        return {"error": "time_reference.py failed: %s" % (e.stderr or "")[-300:]}
    except Exception as e:                                  # noqa: BLE001 -- a reported absence, never a failed bench
        return {"error": "%s: %s" % (type(e).__name__, str(e)[:300])}


def featured_legs(ctx, n_design, base_sw, base_ms, base_pair_iters):
    """The lean featured specialisations of the fused kernel (two waves per SIMD) on the C3 designs, resident in / resident
    out: kernel time per (pair, iteration) against the plain sweep's, and a sample of each leg against the CPU oracle.
    Legs: frequency-dependent M(w), B(w) (the turbine's aerodynamic matrices of raft_model.py:1006-1007: the kernel re-reads
    2 x 36 x nw doubles per pair and iteration -- the leg where HBM traffic starts to matter); C2's three sea states as
    cases; two wave headings (raft_model.py:1200-1236); MacCamy-Fuchs columns (raft_member.py:1415-1420)."""
    from raft_amd import waves
    from raft_amd._abi import RaftxLib
    from raft_amd.metrics import group_rel_err
    nw = base_sw.nw
    w = base_sw.w
    dw = float(w[1] - w[0])
    rng = np.random.default_rng(5)
    sea = lambda Hs, Tp: np.sqrt(2.0 * waves.jonswap(w, Hs, Tp) * dw)
    legs = {}
    specs = {
        "freq_dependent_MB": dict(MBw="aero"),
        "three_sea_states": dict(zeta=np.stack([sea(6.0, 12.0), sea(4.0, 10.0), sea(8.0, 14.0)])[:, None, :], beta=np.zeros((3, 1))),
        "two_headings": dict(zeta=np.stack([sea(6.0, 12.0), sea(3.0, 9.0)])[None, :, :], beta=np.array([[0.0, np.deg2rad(30.0)]])),
        "maccamy_fuchs_columns": dict(mcf=True),
    }
    oracle = RaftxLib(os.path.join(ROOT, "oracle", "libraftx_oracle.so"))
    n_chk = min(256, n_design)                            # designs of every leg re-solved by the oracle's own chain
    for name, kw in specs.items():
        kw = dict(kw)
        if kw.get("MBw") == "aero":                       # smooth in w, different per design: rotor added mass / damping shaped
            amp = rng.uniform(0.5, 1.5, size=(n_design, 2, 6, 6, 1))
            shape = np.stack([1.0 / (1.0 + (w / 0.6) ** 2), (w / 0.8) / (1.0 + (w / 0.8) ** 2)])[None, :, None, None, :]
            kw["MBw"] = np.ascontiguousarray(amp * shape * np.array([2e5, 4e5])[None, :, None, None, None]
                                             * np.eye(6)[None, None, :, :, None])
        sw, _, _ = make_sweep(ctx, n_design, 0, pinned=False, **kw)
        sw.upload(ctx)
        ks = []
        for i in range(6):
            ctx.solve_dynamics_device(sw.nIter, sw.tol, sw.XiStart)
            if i >= 2:
                ks.append(ctx.last_kernel_ms())
        flags, waves_per_simd, slots = ctx.last_solve_kernel()
        r = ctx.fetch_results(want_Xi=True)
        k_ms = float(np.mean(ks))
        pairs = sw.n_design * sw.n_case
        # an extra heading costs one inertial + one drag sweep + the solves of an iteration's worth: count it as one
        pair_iters = float(np.sum(r["niter"])) + pairs * (sw.n_head - 1)
        leg = {"kernel_ms": k_ms, "pairs": int(pairs), "headings": int(sw.n_head), "mean_iterations": float(np.mean(r["niter"])),
               "kernel_flags": int(flags), "waves_per_simd": int(waves_per_simd), "run_start_cache_slots": int(slots),
               "ns_per_pair_iteration": 1e6 * k_ms / pair_iters,
               "vs_plain_sweep": (k_ms / pair_iters) / (base_ms / base_pair_iters),
               "dcf_per_s": pairs * sw.n_head * nw / (k_ms * 1e-3)}
        if sw.MBw is not None:
            A_full = algorithmic_bytes(sw.off, nw) + float(np.sum(r["niter"])) * 2 * 36 * nw * 8.0
            leg["hbm_frac_incl_MB_rereads"] = A_full / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBS
        # the first n_chk designs through the oracle's own chain (geometry, MacCamy-Fuchs tables, fixed point)
        sub = sw.take(0, n_chk)
        o = oracle.context(0)
        sub.upload(o)
        o.solve_dynamics_device(sw.nIter, sw.tol, sw.XiStart)
        ro = o.fetch_results(want_Xi=True)
        o.close()
        err = max(group_rel_err(r["Xi"][d], ro["Xi"][d]) for d in range(n_chk))
        leg["oracle_checked_designs"] = n_chk
        leg["max_group_rel_err_vs_oracle"] = float(err)
        leg["niter_mismatches_vs_oracle"] = int(np.count_nonzero(ro["niter"] != r["niter"][:n_chk]))
        assert err < 1e-6 and leg["niter_mismatches_vs_oracle"] == 0, "featured leg %s fails parity: %r" % (name, leg)
        legs[name] = leg
        del sw, r
    return legs


def emit(out):
    """The ONE JSON line, as the LAST line of stdout: whatever native libraries left in C's stdio buffer (RCCL prints a
    version banner at communicator creation, which a pipe keeps buffered until exit) goes out first."""
    import ctypes
    try:
        ctypes.CDLL(None).fflush(None)
    except OSError:
        pass
    sys.stdout.write(json.dumps(out) + "\n")
    sys.stdout.flush()


def run_streamed(sw_, ctx, n, depth, fresh=None, after=None, chunks=1):
    """n streamed solver-stage steps of the sweep sw_ through the library's slots: depth 2 = submit(i+1), wait(i); depth d > 2 =
    prepare(i+d-1), launch(i+d-2), wait(i).  fresh(): new candidates before a batch is prepared; after(result): per-step
    exchange (the gather).  Returns the per-step results."""
    out_ = []
    def sub(i, submit=False):
        if fresh is not None:
            fresh()
        return (sw_.submit_crossing if submit else sw_.prepare_crossing)(ctx, i % max(depth, 2), n_chunk=chunks)
    def done(r):
        return after(r) if after is not None else r
    if depth <= 2:
        h = sub(0, True) if n > 0 else None
        for i in range(n):
            hn = sub(i + 1, True) if i + 1 < n else None
            out_.append(done(sw_.wait_crossing(ctx, h)))
            h = hn
        return out_
    d = depth
    hs = {i: sub(i) for i in range(min(n, d - 1))}
    for i in range(min(n, d - 2)):
        sw_.launch_crossing(ctx, hs[i])
    for i in range(n):
        if i + d - 1 < n:
            hs[i + d - 1] = sub(i + d - 1)
        if i + d - 2 < n:
            sw_.launch_crossing(ctx, hs[i + d - 2])
        out_.append(done(sw_.wait_crossing(ctx, hs.pop(i))))
    return out_


def guarded(name, fn):
    """A leg OUTSIDE the headline must not cost the run its JSON line: its failure -- a parity assertion included -- is
    reported under the leg's key (and on stderr), the headline and its own all-design parity check stay fatal."""
    try:
        return fn()
    except Exception as e:                                  # noqa: BLE001
        import traceback
        traceback.print_exc()
        return {"error": "%s: %s" % (type(e).__name__, str(e)[:400]), "leg": name}


def launch_ranks(n, argv):
    """`python bench.py --gpus N` without a launcher: start the N ranks (one process per GPU) and wait for them.  Rank r gets
    RANK = LOCAL_RANK = r, WORLD_SIZE = N, a loopback rendezvous on two free ports and a random job token; rank 0 inherits
    stdout (it prints the JSON line), every rank inherits stderr.  A rank that fails takes the others down (by PID) and
    its exit code becomes ours.  RAFTX_BENCH_RANK_CMD (tests): the command to run instead of this file."""
    import secrets
    import shlex
    import socket
    import subprocess
    from raft_amd import backend
    rehearsal = "RAFTX_BENCH_DEVICE" in os.environ
    if "RAFTX_BENCH_RANK_CMD" not in os.environ:
        have = backend.hip_library().device_count()
        if have < n and not rehearsal:
            sys.stderr.write("bench.py: --gpus %d but this host has %d GPU(s) visible (RAFTX_BENCH_DEVICE=<id> rehearses the "
                             "multi-rank path with every rank on one device)\n" % (n, have))
            return 2
    ports = []
    for _ in range(2):                                    # MASTER_PORT (kept for launcher compatibility) and the comm's own port
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            ports.append(sk.getsockname()[1])
    base = dict(os.environ, WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(ports[0]), RAFTX_COMM_PORT=str(ports[1]),
                RAFTX_COMM_TOKEN=os.environ.get("RAFTX_COMM_TOKEN") or secrets.token_hex(16))
    cmd = shlex.split(os.environ["RAFTX_BENCH_RANK_CMD"]) if "RAFTX_BENCH_RANK_CMD" in os.environ else [sys.executable, os.path.abspath(__file__)]
    procs = []
    for r in range(n):
        env = dict(base, RANK=str(r), LOCAL_RANK=str(r))
        procs.append(subprocess.Popen(cmd + list(argv), env=env, stdout=None if r == 0 else subprocess.DEVNULL))
    rc = 0
    live = set(range(n))
    while live:
        for r in sorted(live):
            code = procs[r].poll()
            if code is None:
                continue
            live.discard(r)
            if code != 0 and rc == 0:
                rc = code
                sys.stderr.write("bench.py: rank %d exited with code %d; stopping the other ranks\n" % (r, code))
                for o in live:
                    procs[o].terminate()
        if live:
            time.sleep(0.02)
    return rc


def comm_fallback(rehearsal):
    """What the ranks do when the RCCL communicator cannot be created (or its probe collective fails): "host" = all
    ranks move the exchange step onto the rendezvous channel together and the JSON line's `gather` says so, with the
    reason; "error" = the bench fails.  The sweep has no data-path collective (the gather carries 56 B per
    design-case), so a measured line that names its transport is worth more than no line: "host" unless
    RAFTX_BENCH_COMM_FALLBACK=error."""
    return "host" if rehearsal else os.environ.get("RAFTX_BENCH_COMM_FALLBACK", "host")


def main_sharded_leg(args, world, rank, local, rehearsal):
    """--workload c4 | c5: the sharded forms of configs[3] / configs[4] as the timed workload (bench_legs.py)."""
    import bench_legs
    from raft_amd import backend, comm as rcomm
    ctx = backend.hip_library().context(local)
    comm = gather_kind = None
    if world > 1:
        comm, gather_kind = rcomm.from_env(ctx, prefer="rccl", fallback=comm_fallback(rehearsal))
    ctx.synchronize()
    if comm is not None:
        comm.barrier()
    if args.workload == "c4":
        leg = bench_legs.c4_farm(ctx, farms=args.farms, repeat=max(args.steps, 2), comm=comm)
        out = None
        if rank == 0:
            fs = leg["farm_sweep"]
            out = {"metric": "design-case-frequency solves/sec (whole node)", "unit": "dcf solves/s",
                   "value": world * fs["dcf_per_s_kernels"], "n_gpus": world, "steps": args.steps, "warmup": 1,
                   "ms_per_step": fs["unit_fixed_points_kernel_ms"] + fs["coupled_solves_kernel_ms"], "higher_is_better": True,
                   "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
                   "config": {"workload": "C4 farm sweep: %d farms/GPU x 4 units x 50 sea states x 200 bins (unit fixed points + coupled 24 x 24 "
                                          "solves, kernels of rank 0); the farm as specified sharded by sea state" % args.farms, "gather": gather_kind},
                   "roofline": fs["roofline"], "c4_farm": leg}
    else:
        t0 = time.perf_counter()
        b = bench_legs.c5_qtf_batch(ctx, n_set=args.sets, repeat=max(args.steps, 2), deck="c5_oc4semi_qtf.npz", comm=comm, kay=True)
        dt = b["wall_ms"] * 1e-3
        times = [b["qtf_kernels_ms"], b["wall_ms"]]
        allt = comm.gather_floats(times) if comm is not None else np.array([times])
        out = None
        if rank == 0:
            q = b["q"]
            herm = bool(np.allclose(q[0], np.conj(np.transpose(q[0], (1, 0, 2))), atol=1e-6 * np.abs(q[0]).max()))
            assert herm and np.all(np.isfinite(q.view(float))), "reduced QTF is not Hermitian / finite"
            k_max = float(np.max(allt[:, 0]))
            out = {"metric": "QTF difference-frequency pairs/sec (whole node)", "unit": "pairs/s", "value": b["pairs"] / dt,
                   "n_gpus": world, "steps": max(args.steps, 2), "warmup": 1, "ms_per_step": b["wall_ms"], "higher_is_better": True,
                   "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
                   "config": {"workload": "C5 OC4semi-RAFT_QTF: %d sets x 200 x 200 grid (20 100 pairs each), rows interleaved over ranks, one "
                                          "SUM-reduce onto rank 0; step = Kim & Yue tables + QTF kernels + download + reduce (wall of rank 0)" % args.sets,
                              "gather": gather_kind},
                   "roofline": bench_legs._roof(b["pairs"] * b["strips"] * bench_legs.QTF_FLOP_PER_STRIP_PAIR / world, k_max,
                                                "k_qtf_pairs (+ k_qtf_tables), slowest rank"),
                   "per_rank_ms": {"qtf_kernels": [float(x) for x in allt[:, 0]], "wall": [float(x) for x in allt[:, 1]]},
                   "hermitian": herm}
    if comm is not None:
        comm.close()
    ctx.close()
    if rank == 0:
        emit(out)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--workload", choices=("c3", "c4", "c5"), default="c3", help="c3: the headline design sweep (default); c4 / c5: the sharded "
                                                                                "farm / QTF workloads of BASELINE configs[3] / [4]")
    ap.add_argument("--farms", type=int, default=200, help="--workload c4: farms per GPU of the farm sweep")
    ap.add_argument("--sets", type=int, default=16, help="--workload c5: QTF sets per batch")
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--designs", type=int, default=10000, help="designs per GPU (weak scaling) / designs of the whole sweep (strong scaling)")
    ap.add_argument("--descriptors", choices=("device", "host"), default="device",
                    help="device (default): every step solves NEW candidates -- their five parameters go in, the library writes the "
                         "member descriptors in HBM (raftx_sweep_prepare_variants).  host: round 1-4's form -- the descriptors of ONE "
                         "batch are expanded by NumPy once (geometry.host_descriptor_ms) and the same 66 MB are re-uploaded every step")
    ap.add_argument("--scaling", choices=("weak", "strong"), default="weak",
                    help="weak (default, what the driver's contract line reports): --designs per GPU.  strong: ONE sweep of --designs "
                         "(BASELINE configs[2]: 10 000) cut into N contiguous shards, SURVEY 8e (1 250 per GPU at N = 8)")
    ap.add_argument("--chunks", type=int, default=0, help="design blocks per step (0 = library default)")
    ap.add_argument("--workers", type=int, default=0, help="internal streams (0 = library default)")
    ap.add_argument("--xi-out", action="store_true", help="download the full responses inside the step as well")
    ap.add_argument("--pageable", action="store_true", help="descriptors in ordinary NumPy memory instead of page-locked")
    ap.add_argument("--resident", action="store_true", help="also time the fused kernel alone, whole batch resident in HBM")
    ap.add_argument("--no-stream", action="store_true", help="time isolated blocking calls (raftx_sweep_stats) instead of streaming the "
                                                             "steps through the library's slots (raftx_sweep_prepare / _launch / _wait)")
    ap.add_argument("--no-cpu-baseline", action="store_true", help="skip the oracle leg (cpu_baseline and the all-design check)")
    ap.add_argument("--depth", type=int, default=0, help="batches in flight when the steps are streamed: 2 = submit(i+1), wait(i) (default); 4 = prepare(i+3), launch(i+2), wait(i) "
                                                      "(default with --xi-out: the download of a batch takes longer than its kernels); "
                                                      "3 = prepare(i+2), launch(i+1), wait(i): the next batch's tables are generated in the drain "
                                                      "of the running fused kernel -- measured on one box over 300 steps: 3.73 against 3.71 ms per step "
                                                      "(the gap between fused kernels shrinks 0.35 -> 0.2 ms, the kernels sharing the drain slow each "
                                                      "other by as much), so it is not the default")
    ap.add_argument("--no-extra-legs", action="store_true", help="skip the legs outside the headline (xi-out, featured sweeps)")
    ap.add_argument("--legs", default="xi,featured,configs,hostdesc,traffic,launchsize,shard", help="which legs outside the headline run (comma list of xi, featured, configs, hostdesc, launchsize, shard, "
                                                                            "traffic = the FETCH_SIZE / WRITE_SIZE passes of roofline.traffic, two child runs under rocprofv3)")
    ap.add_argument("--profile", action="store_true", help="for runs under rocprofv3: whole-batch launches only (--chunks 1), no isolated / "
                                                            "extra / oracle legs -- ONE population of k_solve_dynamics launches in the trace")
    args = ap.parse_args()
    if args.profile:
        args.chunks, args.no_extra_legs, args.no_cpu_baseline = 1, True, True
    # NumPy's BLAS pool on at most eight threads for the life of this process (it only ever lowers the count): the pool's GPU
    # boxes give a process 16 CPUs by cgroup quota behind 256 logical ones, and OpenBLAS's idle-spinning threads spend that
    # quota -- the whole process, the thread that feeds the GPU included, is then throttled (raft_amd/hostblas.py)
    global _BLAS_SCOPE
    from raft_amd import hostblas
    _BLAS_SCOPE = hostblas.few_threads()
    _BLAS_SCOPE.__enter__()
    if args.depth == 0:
        # default: two batches in flight with host-made descriptors (their upload rides the DMA engines); three, staged, with
        # device-made ones -- the expansion of batch i+2 then has a whole step to trickle in beside the fused kernels instead
        # of holding up the member pass of the very next batch (same box, K = 40, gpurun_out/r05_depth: 3.22 / 3.17 / 3.16 ms
        # at depth 2 / 3 / 4; host-made: 3.08 / 3.09 / 3.16); four when the responses are downloaded
        args.depth = 4 if args.xi_out else (3 if args.descriptors == "device" else 2)

    if args.gpus < 1:
        ap.error("--gpus must be >= 1")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:  # no launcher around us: be the launcher
        sys.exit(launch_ranks(args.gpus, sys.argv[1:]))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if rank != 0:                                         # stdout carries ONE line, rank 0's: whatever native libraries of the other
        dn = os.open(os.devnull, os.O_WRONLY)             # ranks print there (RCCL's version banner at exit) must not follow it
        os.dup2(dn, 1)
        os.close(dn)
    if world != args.gpus:                                # a line that says n_gpus = N must have been produced by N ranks
        sys.stderr.write("bench.py: --gpus %d but WORLD_SIZE=%d: start the ranks with --nproc-per-node %d (or drop the launcher: "
                         "`python bench.py --gpus %d` starts them itself)\n" % (args.gpus, world, args.gpus, args.gpus))
        sys.exit(2)
    # RAFTX_BENCH_DEVICE=0: rehearsal of the multi-rank path on a single-GPU box (all ranks share device 0; RCCL refuses
    # that, so the exchange steps fall back to the host transport and the JSON line says so); the driver's runs have one
    # GPU per rank and use RCCL or fail
    rehearsal = "RAFTX_BENCH_DEVICE" in os.environ
    if rehearsal:
        local = int(os.environ["RAFTX_BENCH_DEVICE"])

    if args.workload != "c3":
        return main_sharded_leg(args, world, rank, local, rehearsal)

    from raft_amd import backend
    from raft_amd.metrics import rao_group_err
    # the rank's host thread and page-locked buffers go onto the socket its GPU hangs off (raft_amd/locality.py);
    # the original CPU set comes back for the oracle leg, which wants every core
    from raft_amd import locality
    cpus_at_start = os.sched_getaffinity(0)
    placement = ({"bound": False, "why": "RAFTX_NO_BIND"} if os.environ.get("RAFTX_NO_BIND")
                 else locality.bind_near_device(backend.hip_library(), local))
    ctx = backend.hip_library().context(local)
    from raft_amd.sweep import shard_bounds
    strong = args.scaling == "strong"
    shard = [shard_bounds(args.designs, r_, world) for r_ in range(world)] if strong else [(r_ * args.designs, (r_ + 1) * args.designs) for r_ in range(world)]
    counts_all = np.array([hi_ - lo_ for lo_, hi_ in shard], dtype=np.int64)
    if strong and counts_all.min() < 1:
        sys.stderr.write("bench.py: --scaling strong needs at least one design per rank\n")
        sys.exit(2)
    variants = args.descriptors == "device"
    sw, fx, geo = make_sweep(ctx, args.designs, rank, pinned=not args.pageable, rows=shard[rank], variants=variants)
    nw, nD = sw.nw, sw.n_design
    # distinct candidates per step: batch b >= 1 of this rank = rows shard + b * (all ranks' designs) of the same stream
    # (batch 0 = the standard rows, whose first 64 designs are the reference-built variants: the parity batch)
    from raft_amd import geometry as G_
    n_all = int(counts_all.sum()) if strong else args.designs * world
    batch_no = {"next": 1}
    host_params_s = []

    def fresh_candidates():
        if not variants:
            return
        tq = time.perf_counter()
        b = batch_no["next"]
        batch_no["next"] += 1
        sw.set_params(G_.volturnus_params(scale_rows(shard[rank][0] + b * n_all, shard[rank][1] + b * n_all)))
        host_params_s.append(time.perf_counter() - tq)

    comm = None
    gather_kind = None
    ctx_comm = None
    if world > 1:
        from raft_amd import comm as rcomm
        # the exchange step gets its own context (= its own stream): the gather of step i must not queue behind the
        # kernels of step i + 1, which are already on the solver context's stream when the steps are streamed
        ctx_comm = backend.hip_library().context(local)
        comm, gather_kind = rcomm.from_env(ctx_comm, prefer="rccl", fallback=comm_fallback(rehearsal))

    stream_steps = not args.no_stream
    Xi_pinned = [ctx.pinned_empty((nD, 1, 1, 6, nw)) for _ in range(4 if stream_steps else 1)] if args.xi_out else [None, None, None, None]

    gather_s = []                                         # host time of every gather (this rank's share of the exchange step)
    solo = {"on": False}                                  # rank 0's single-rank leg runs the same steps without the exchange

    def gather(r):
        if comm is not None and not solo["on"]:           # statistics of every rank onto rank 0 (48 B + 8 B per design-case)
            tg = time.perf_counter()
            r["std_all"] = comm.gather_rows(np.concatenate([r["std"].reshape(nD, -1), r["niter"].reshape(nD, -1).astype(np.float64)], axis=1),
                                            counts=counts_all)            # weak: nD on every rank; strong: the shards' sizes
            gather_s.append(time.perf_counter() - tg)
        return r

    def step():                                           # one isolated, blocking crossing
        fresh_candidates()
        return gather(sw.run_crossing(ctx, n_chunk=args.chunks, n_worker=args.workers, Xi_out=Xi_pinned[0]))

    def run_steps(n):
        """n whole solver-stage steps; returns the per-step results.  Streamed: step i + 1 is submitted (its descriptor
        upload and member pass start) before step i is collected, as consecutive batches of a long sweep are; every
        step still moves its own descriptors in and its own statistics out."""
        if not stream_steps:
            return [step() for _ in range(n)]
        out = []
        def submit(slot):
            fresh_candidates()                            # (the handle keeps its parameter rows alive until it is waited for)
            return sw.submit_crossing(ctx, slot, n_chunk=args.chunks, Xi_out=Xi_pinned[slot])
        if args.depth == 2:
            h = submit(0) if n > 0 else None
            for i in range(n):
                h_next = submit((i + 1) % 2) if i + 1 < n else None
                out.append(gather(sw.wait_crossing(ctx, h)))
                h = h_next
            return out
        d = args.depth                                    # staged: prepare(i+d-1), launch(i+d-2), wait(i)

        def sub(i):
            fresh_candidates()
            return sw.prepare_crossing(ctx, i % d, n_chunk=args.chunks, Xi_out=Xi_pinned[i % d])
        hs = {i: sub(i) for i in range(min(n, d - 1))}
        for i in range(min(n, d - 2)):
            sw.launch_crossing(ctx, hs[i])
        for i in range(n):
            if i + d - 1 < n:
                hs[i + d - 1] = sub(i + d - 1)
            if i + d - 2 < n:
                sw.launch_crossing(ctx, hs[i + d - 2])
            out.append(gather(sw.wait_crossing(ctx, hs.pop(i))))
            if os.environ.get("RAFTX_BENCH_DEBUG"):
                print("  step %d collected at %.3f ms" % (i, 1e3 * time.perf_counter()), file=sys.stderr)
        return out

    def barrier():                                        # device idle on every rank, then all ranks together, both sides
        ctx.synchronize()
        if comm is not None:
            ctx_comm.synchronize()
            comm.barrier()

    if stream_steps:
        # untimed priming, before the W warm-up steps: the first crossing of a stream is cut into two blocks, the following
        # ones are single blocks, and each slot's block contexts allocate their device buffers the first time they meet a
        # configuration -- three streamed steps bring both slots to the steady-state one (seven the three of --depth 3).
        # And the chip's clocks: a kernel trace of this loop (profiles/r05_final/kernel_stats.csv, gpurun_out/r05_final) shows
        # the first fused launches of a process at 3.2, 3.1, 3.06, 3.03, 2.99, 2.92, 2.87, 2.85 ms before they settle at
        # 2.76-2.80 -- ~9 launches of power-management ramp that a long sweep sees once.  Twelve priming steps (36 ms) put the
        # W warm-up steps and the K timed ones behind it, whatever W the caller chose.
        run_steps(max(12, 2 * args.depth + 1))
    run_steps(args.warmup)
    # N > 1: the single-rank yardstick of THIS invocation -- rank 0 alone runs the same K steps (no exchange step) while the
    # other ranks wait at the barrier, so that the N-rank value can be set against N x one rank on the same box and build
    single_rank = None
    if comm is not None:
        barrier()
        if rank == 0:
            solo["on"] = True
            ctx.synchronize()
            ts = time.perf_counter()
            run_steps(args.steps)
            ctx.synchronize()
            single_rank = (time.perf_counter() - ts) / args.steps
            solo["on"] = False
    # N > 1, weak scaling (the contract line): ALSO the literal shape of BASELINE configs[2] in the same invocation -- ONE sweep of
    # --designs cut into N contiguous shards, streamed and gathered like the timed steps
    strong_same = None
    if comm is not None and not strong and args.designs >= world:
        sb_ = [shard_bounds(args.designs, r_, world) for r_ in range(world)]
        cnt_ = np.array([hi_ - lo_ for lo_, hi_ in sb_], dtype=np.int64)
        sw_st, _, _ = make_sweep(ctx, int(cnt_[rank]), rank, pinned=not args.pageable, rows=sb_[rank], variants=variants)
        bno_ = {"next": 1}

        def fresh_st():
            if variants:
                b = bno_["next"]
                bno_["next"] += 1
                sw_st.set_params(G_.volturnus_params(scale_rows(sb_[rank][0] + b * args.designs, sb_[rank][1] + b * args.designs)))

        def gather_st(r):
            n_ = int(cnt_[rank])
            comm.gather_rows(np.concatenate([r["std"].reshape(n_, -1), r["niter"].reshape(n_, -1).astype(np.float64)], axis=1), counts=cnt_)
            return r
        run_streamed(sw_st, ctx, max(6, 2 * args.depth + 1), args.depth, fresh=fresh_st, after=gather_st)
        barrier()
        ts_ = time.perf_counter()
        run_streamed(sw_st, ctx, args.steps, args.depth, fresh=fresh_st, after=gather_st)
        barrier()
        el_ = comm.all_max(time.perf_counter() - ts_)
        strong_same = {"scaling": "strong", "total_designs": int(args.designs), "shard_designs": [int(c_) for c_ in cnt_],
                       "ms_per_step": 1e3 * el_ / args.steps, "value": args.designs * nw * args.steps / el_, "steps": args.steps,
                       "note": "BASELINE configs[2] literally: ONE sweep of %d designs cut into %d shards (one per GPU), each step = every rank's "
                               "shard solved and its statistics gathered on rank 0; same ranks, same invocation as the weak-scaling line above"
                               % (args.designs, world)}
        del sw_st
    barrier()
    del gather_s[:]
    del host_params_s[:]
    first_timed_batch = batch_no["next"]
    t0 = time.perf_counter()
    res = run_steps(args.steps)                           # returns after the last step's streams have drained
    t_own = time.perf_counter() - t0                      # this rank's own K steps, before it waits for the others
    barrier()
    elapsed = time.perf_counter() - t0
    r = res[-1]
    tims = [x["timing_ms"] for x in res]
    host_params_ms = 1e3 * float(np.mean(host_params_s)) if host_params_s else None
    isolated = None
    if stream_steps and rank == 0 and not args.profile:   # the same step as an isolated blocking call (outside the timed region)
        t1 = time.perf_counter()
        for _ in range(5):                                # (no gather here: only this rank runs it)
            fresh_candidates()
            sw.run_crossing(ctx, n_chunk=args.chunks, n_worker=args.workers, Xi_out=Xi_pinned[0])
        isolated = (time.perf_counter() - t1) / 5
    per_rank = None
    if comm is not None:
        elapsed = comm.all_max(elapsed)                   # the slowest rank's clock
        per_rank = comm.gather_floats([1e3 * t_own / args.steps, 1e3 * float(np.mean(gather_s)) if gather_s else 0.0,
                                       float(geo["host_descriptor_ms"])])
    tims = np.array(tims)
    nan = int(sum(np.count_nonzero(x["flags"] & 2) for x in res))
    # algorithmic work of the TIMED steps (each step its own candidates when the descriptors are device-made): means per step
    A_steps = float(np.mean([algorithmic_bytes(x["strip_off"], nw) for x in res]))
    flops_steps = float(np.mean([algorithmic_flops(x["strip_off"], nw, x["niter"]) for x in res]))
    mean_iter_timed = float(np.mean([np.mean(x["niter"]) for x in res]))
    if comm is not None and rank == 0:
        assert r["std_all"].shape == (int(counts_all.sum()), 7) and np.array_equal(r["std_all"][:nD, :6], r["std"].reshape(nD, 6))

    # ---- outside the timed region: determinism of a timed batch, then parity on the STANDARD batch (rows whose first 64
    # designs the live reference solved), one more crossing with the responses downloaded
    if variants:
        distinct = len({x["std"].tobytes() for x in res})
        assert distinct == len(res), "timed steps were meant to solve distinct candidates (%d distinct of %d)" % (distinct, len(res))
        b_last = first_timed_batch + len(res) - 1             # the last timed step's candidates, once more, alone
        sw.set_params(G_.volturnus_params(scale_rows(shard[rank][0] + b_last * n_all, shard[rank][1] + b_last * n_all)))
        again = sw.run_crossing(ctx, n_chunk=args.chunks, n_worker=args.workers)
        assert np.array_equal(again["std"].view(np.uint64), r["std"].view(np.uint64)) and np.array_equal(again["niter"], r["niter"]), \
            "the same candidates solved twice (streamed / alone) differ"
        sw.set_params(G_.volturnus_params(scale_rows(*shard[rank])))
    # (--profile: no download -- a crossing that downloads its responses is cut into slabs of one residency round, ten more
    # launches of the same kernel name in the rocprofv3 statistics; the parity of the responses is the default run's business)
    chk = sw.run_crossing(ctx, n_chunk=args.chunks, n_worker=args.workers, want_Xi=not args.profile)
    if not variants:
        for x in res:                                     # every timed step re-solved the same batch: the same bits
            assert np.array_equal(x["std"].view(np.uint64), r["std"].view(np.uint64)) and np.array_equal(x["niter"], r["niter"])
        assert np.array_equal(chk["std"].view(np.uint64), r["std"].view(np.uint64)) and np.array_equal(chk["niter"], r["niter"]), \
            "two crossings of the same batch differ"
    off = chk["strip_off"]
    niter = chk["niter"]
    nan += int(np.count_nonzero(chk["flags"] & 2))
    Xi = chk["Xi"]
    parity = {"nan_flags": nan}
    if rank == 0 and Xi is not None:                      # the first 64 designs are the live reference's own variants
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
        cpu, ores, gen_chk = oracle_run(sw, ctx)
        parity["generator_vs_oracle_all_designs"] = gen_chk
        assert gen_chk["strip_count_mismatches"] == 0 and gen_chk.get("strips_max_err_rel_to_field_max", 1.0) < 1e-9 and \
            gen_chk.get("strip_index_mismatches", 1) == 0 and \
            all(v < 1e-7 for k_, v in gen_chk.items() if k_.endswith("_max_group_rel_err")), \
            "device-generated tables / statics differ from the oracle's generator: %r" % gen_chk
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

    # ---- legs outside the headline (N = 1): SURVEY 8d's literal step (responses downloaded), featured sweeps
    xi_leg = featured = None
    import types
    import bench_legs
    B = types.SimpleNamespace(ctx=ctx, sw=sw, nD=nD, nw=nw, args=args, Xi=Xi, chk=chk, variants=variants, rank=rank, shard=shard,
                              make_sweep=make_sweep, run_streamed=run_streamed, algorithmic_flops=algorithmic_flops, scale_rows=scale_rows,
                              G_=G_)                     # what the legs in bench_legs.py read of this run
    if rank == 0 and world == 1 and not args.no_extra_legs and not args.xi_out and "xi" in args.legs:
        xi_leg = guarded("xi_out", lambda: bench_legs.xi_out(B))
        if "error" in xi_leg:                             # a crossing may still be in flight: drain before the next leg
            try:
                ctx.synchronize()
            except Exception:                             # noqa: BLE001
                pass
    def run_featured():
        sw.upload(ctx)                                    # the plain sweep, resident: the yardstick of the featured legs
        ks = []
        for i in range(6):
            ctx.solve_dynamics_device(sw.nIter, sw.tol, sw.XiStart)
            if i >= 2:
                ks.append(ctx.last_kernel_ms())
        f_ = {"plain_sweep": {"kernel_ms": float(np.mean(ks)), "pairs": int(nD), "kernel_flags": ctx.last_solve_kernel()[0],
                                    "ns_per_pair_iteration": 1e6 * float(np.mean(ks)) / float(np.sum(niter))}}
        f_.update(featured_legs(ctx, nD, sw, float(np.mean(ks)), float(np.sum(niter))))
        return f_
    if rank == 0 and world == 1 and not args.no_extra_legs and not args.xi_out and "featured" in args.legs:
        featured = guarded("featured_sweeps", run_featured)
    # ---- the rounds-1-4 form of the step on the same box, for comparison: ONE batch expanded by NumPy outside the step and its
    # 66 MB of descriptors re-uploaded (DMA) every step -- no device-side expansion, but a host that can feed only 1 / 14 ms
    hostdesc = None

    if variants and rank == 0 and world == 1 and not args.no_extra_legs and not args.xi_out and "hostdesc" in args.legs:
        hostdesc = guarded("host_descriptors_same_box", lambda: bench_legs.host_descriptors(B))

    # ---- BASELINE configs[1], [3], [4] at their specified sizes, each against its live-reference golden (N = 1)
    cfg_legs = {}
    if rank == 0 and world == 1 and not args.no_extra_legs and "configs" in args.legs:
        import bench_legs
        cfg_legs["c2_dropin"] = guarded("c2_dropin", lambda: bench_legs.c2_dropin(ctx))
        cfg_legs["c4_farm"] = guarded("c4_farm", lambda: bench_legs.c4_farm(ctx, farms=1000))
        cfg_legs["c5_qtf"] = guarded("c5_qtf", lambda: bench_legs.c5_qtf(ctx))
        cfg_legs["flex_sweep"] = guarded("flex_sweep", lambda: bench_legs.flex_sweep(ctx))

    # ---- the fused kernel against the size of its launch (N = 1): the same designs' stream, 20 000 and 40 000 pairs resident.
    # BASELINE's configuration is 10 000 pairs per GPU = 9.77 residency rounds: the last one drains with nothing left to hand out
    # (914 of 1 024 slots busy on average; DESIGN.md 3.1, profiles/r05_launch_size_scaling.json); these two figures say what the
    # kernel does when a launch is long enough for that not to matter.
    launch_size = None
    if rank == 0 and world == 1 and not args.no_extra_legs and "launchsize" in args.legs:
        launch_size = guarded("launch_size", lambda: bench_legs.launch_size(B))

    # ---- the shard of BASELINE configs[2]'s strong-scaling shape on ONE GPU: 10 000 designs over 8 ranks = 1 250 per rank
    # (raft/parametersweep.py:39-100).  The same streamed step at 1 250 designs: what one rank of an 8-GPU
    # strong-scaling run does, so its ms_per_step against this run's 10 000-design step IS the projected 8-GPU speed-up
    # (no collective while solving; the gather moves 56 B per design).
    shard_leg = None
    if rank == 0 and world == 1 and not args.no_extra_legs and stream_steps and "shard" in args.legs and nD >= 8:
        shard_leg = guarded("shard_1250", lambda: bench_legs.shard_1250(B, 1e3 * elapsed / args.steps))

    # ---- roofline.traffic: measured in THIS run where rocprofv3 is at hand (N = 1), else the committed profile's figure
    traffic_bytes, traffic_prov = measured_traffic(nD), traffic_provenance()
    if rank == 0 and world == 1 and not args.no_extra_legs and not args.profile and "traffic" in args.legs:
        tb, tp = live_traffic(nD)
        if tb is not None:
            traffic_bytes, traffic_prov = tb, tp
        else:
            traffic_prov = dict(traffic_prov or {}, live_measurement="not made: %s" % tp)

    n_dcf_rank = nD * 1 * nw
    value = int(counts_all.sum()) * nw * args.steps / elapsed
    k_each_ms = float(np.mean(tims[:, 2]))                # k_solve_dynamics: HIP-event duration of each step's launch(es), start to end
    # The fused kernels of consecutive batches OVERLAP (alternating streams: the drain of batch i is the ramp of batch i+1),
    # so the time the chip spends in k_solve_dynamics per step is the UNION of the launches' spans on the device's clock
    # (raftx_sweep_solve_span) over the K timed steps / K -- not the sum of the per-launch durations, which counts every
    # overlap twice.  Without the spans (isolated calls) the per-launch duration stands.
    spans = sorted((float(x["solve_span_ms"][0]), float(x["solve_span_ms"][1])) for x in res
                   if "solve_span_ms" in x and x["solve_span_ms"][1] > x["solve_span_ms"][0])
    k_union_ms = None
    if len(spans) == len(res) and spans:
        busy, (lo_, hi_) = 0.0, spans[0]
        for a_, b_ in spans[1:]:
            if a_ > hi_:
                busy += hi_ - lo_
                lo_, hi_ = a_, b_
            else:
                hi_ = max(hi_, b_)
        busy += hi_ - lo_
        k_union_ms = busy / len(res)
    k_sum_ms = k_union_ms if k_union_ms is not None else k_each_ms
    A = A_steps
    flops = flops_steps
    out = {
        "metric": "design-case-frequency solves/sec (whole node)",
        "value": value, "unit": "dcf solves/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1e3 * elapsed / args.steps,
        "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
        "dtype": "f64", "data": "synthetic",
        "config": {"workload": "C3 VolturnUS-S parameter sweep: %s x 1 sea state (JONSWAP Hs6 Tp12, 0 deg) x %d bins, "
                               "nIter=4, tol=0.01; designs = distinct U[0.75,1.25]^5 variants (default_rng(0))"
                               % (("ONE sweep of %d designs in %d contiguous shards" % (int(counts_all.sum()), world)) if strong
                                  else "%d designs/GPU" % nD, nw),
                   "designs_per_gpu": nD, "shard_designs": [int(c) for c in counts_all], "total_designs": int(counts_all.sum()),
                   "cases": 1, "nw": nw,
                   "step": "whole solver stage (SURVEY 8d): %s + table/statics generation + fused fixed point + "
                           "statistics + D2H of %s; %s"
                           % ("H2D of the candidates' PARAMETERS (5 per design; NEW candidates every step) + member descriptors written "
                              "on the device (k_geom_expand)" if variants else "descriptor H2D (the same batch every step)",
                              "statistics and full responses" if args.xi_out else "statistics (\"stats out\")",
                              "steps streamed through the library's slots (raftx_sweep_submit / raftx_sweep_wait): the upload and "
                              "member pass of step i+1 run beside the kernels of step i, as consecutive batches of a long sweep do; every "
                              "step moves its own inputs in and its own statistics out" if stream_steps
                              else "isolated blocking calls"),
                   "descriptors": args.descriptors,
                   "streamed": bool(stream_steps),
                   "state": "xi out" if args.xi_out else "stats out",
                   "sharding": "designs over ranks, no collective while solving; statistics gathered to rank 0 inside the step"
                               if world > 1 else "single GPU",
                   "gather": gather_kind},
        "step_breakdown_ms": {("latency_submit_to_collected" if stream_steps else "wall_in_library"): float(np.mean(tims[:, 0])),
                              "generation_kernels_sum": float(np.mean(tims[:, 1])),
                              "solve_kernels_sum": k_sum_ms, "statistics_kernels_sum": float(np.mean(tims[:, 3]))},
        "parity": parity,
        "mean_iterations": mean_iter_timed,
        # the BINDING roof: the fused kernel is fp64-VALU-bound (370 FLOP per algorithmic byte against a machine balance of
        # 10); the HBM view and the measured traffic sit beside it
        "roofline": {"bound": "fp64_valu", "achieved": flops / (k_sum_ms * 1e-3) / 1e12, "peak": FP64_VALU_PEAK_TF, "unit": "TFLOP/s",
                     "frac": flops / (k_sum_ms * 1e-3) / 1e12 / FP64_VALU_PEAK_TF,
                     "traffic": traffic_bytes, "traffic_from_profile": traffic_prov,
                     "kernel": "k_solve_dynamics (persistent form raftx_kp_f0: the same body, pairs claimed by a resident grid)", "kernel_ms_per_step": k_sum_ms, "algorithmic_flops_per_step": flops,
                     "kernel_time_is": ("union of the fused launches' spans over the K timed steps / K (consecutive launches overlap)"
                                        if k_union_ms is not None else "HIP events around each launch"),
                     "kernel_ms_per_launch": k_each_ms,
                     "step_frac": flops / (1e-3 * 1e3 * elapsed / args.steps) / 1e12 / FP64_VALU_PEAK_TF,
                     "hbm": {"achieved": A / (k_sum_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                             "frac": A / (k_sum_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "algorithmic_bytes_per_step": A},
                     "sustained_fma_probe": {"tflops": FP64_FMA_SUSTAINED_TF,
                                             "frac": flops / (k_sum_ms * 1e-3) / 1e12 / FP64_FMA_SUSTAINED_TF,
                                             "note": "what a pure v_fma_f64 loop sustains on an MI355X at the kernel's occupancy "
                                                     "(the clock drops to 1.7 GHz under fp64 load): scripts/ubench/valu_mfma_probe.hip, "
                                                     "profiles/r02_valu_mfma_probe.jsonl"},
                     "note": "kernel time = HIP events around every k_solve_dynamics launch of the timed steps, summed per step; "
                             "traffic = 2 x FETCH_SIZE + WRITE_SIZE of separate rocprofv3 --pmc passes: made by this run (traffic_from_profile."
                             "measured_in_this_run) or, where that was not possible, the committed profile's (profiles/traffic_latest.json)"},
        "host_placement": placement,
        "geometry": dict(geo, strips=int(off[-1]), strip_table_bytes_not_uploaded=int(off[-1]) * 256,
                         host_params_ms_per_step=host_params_ms, distinct_candidates_per_step=bool(variants)),
    }
    if per_rank is not None and rank == 0:
        out["per_rank_ms"] = {"min": float(per_rank[:, 0].min()), "max": float(per_rank[:, 0].max()), "all": [float(x) for x in per_rank[:, 0]],
                              "note": "every rank's own K steps / K, before the closing barrier"}
        out["host_descriptor_ms_per_rank"] = {"all": [float(x) for x in per_rank[:, 2]],
                                              "note": "host NumPy time each rank spent expanding ITS designs' descriptors (outside the step): "
                                                      "ranks of one socket contend for it when they all do it at once"}
        out["gather_ms"] = {"rank0": float(per_rank[0, 1]), "max": float(per_rank[:, 1].max()),
                            "note": "host time inside the per-step exchange (%s); on rank 0 it includes waiting for the slowest rank's rows" % gather_kind}
        if single_rank is not None:
            v1 = n_dcf_rank / single_rank
            out["single_rank_same_invocation"] = {"ms_per_step": 1e3 * single_rank, "value": v1, "designs": int(nD),
                                                  "note": "rank 0 alone on ITS %d designs, same K steps, the other ranks idle at the barrier"
                                                          "%s" % (nD, " (strong scaling: a shard, not the whole sweep -- the one-GPU time of the whole "
                                                                      "sweep is the N = 1 run of the same command)" if strong else "")}
            out["scaling_efficiency"] = value / (world * v1)
    if strong_same is not None and rank == 0:
        out["strong_same_invocation"] = strong_same
    if isolated is not None:
        out["isolated_call"] = {"ms_per_step": 1e3 * isolated, "dcf_per_s_per_gpu": n_dcf_rank / isolated,
                                "note": "the same step as one blocking raftx_sweep_stats call with nothing else in flight "
                                        "(upload of all descriptors on the critical path)"}
    if resident is not None:
        out["kernel_resident"] = resident
    if launch_size is not None:
        out["launch_size"] = launch_size
    if shard_leg is not None:
        out["shard_1250"] = shard_leg
    if xi_leg is not None:
        out["xi_out"] = xi_leg
        if isinstance(xi_leg, dict) and "streamed_dcf_per_s" in xi_leg:     # SURVEY 8d's literal step (D2H of Xi), beside `value`
            out["value_xi_out"] = xi_leg["streamed_dcf_per_s"]
    if featured is not None:
        out["featured_sweeps"] = featured
    if hostdesc is not None:
        out["host_descriptors_same_box"] = hostdesc
    out.update(cfg_legs)
    # ---- the CPU baseline (rank 0, N = 1): the UNMODIFIED reference on this host's cores when this host has it (build
    # container: /root/reference; GPU box: oracle/_ref/raft_reference.zip), the vectorised C port beside it
    ref_here = reference_on_this_host() if (rank == 0 and world == 1 and not args.no_cpu_baseline) else None
    if ref_here is not None and "error" not in ref_here:
        nw_ = nw
        out["cpu_baseline"] = {
            "value": ref_here["dcf_per_s_pool"], "unit": "dcf solves/s", "cores": ref_here["pool_procs"], "kind": "reference",
            "implementation": "reference-numpy: the unmodified raft.Model.solveDynamics (raft/raft_model.py:966) under the stub "
                              "recipe of SURVEY.md 8c, imported from %s" % ("oracle/_ref/raft_reference.zip (byte-compiled by "
                              "oracle/stage_reference.py from the sources under /root/reference)" if ref_here["reference_from"] == "archive"
                              else "/root/reference"),
            "sample": "C3 sweep variants (the first 64 designs of the timed batch, cyclically) x 1 sea state x %d bins: "
                      "multiprocessing.Pool(%d) over %d (design, case) items, BLAS/OpenMP threads = 1, %.1f s wall (each item = "
                      "Model(design) + statics + hydro constants + solveDynamics); one core alone: %d designs, %.2f s per solveDynamics; "
                      "every timed solve equals tests/golden/c3_variants.npz (max rel. diff %.1e)"
                      % (nw_, ref_here["pool_procs"], ref_here["pool_items"], ref_here["pool_wall_s"], ref_here["one_core_designs"],
                         ref_here["solveDynamics_s_per_design"], ref_here["max_rel_err_vs_committed_reference_fixture"]),
            "dcf_per_s_one_core": ref_here["dcf_per_s_one_core"], "dcf_per_s_pool": ref_here["dcf_per_s_pool"],
            "dcf_per_s_pool_solve_only": ref_here["dcf_per_s_pool_solve_only"], "cores_on_host": ref_here["cores_on_host"],
            "details": ref_here,
            "gpu_over_reference": {"one_core": value / ref_here["dcf_per_s_one_core"], "pool_all_cores": value / ref_here["dcf_per_s_pool"],
                                   "note": "a reported ratio, not a quality claim: the roofline fraction is"},
        }
        if cpu is not None:
            out["cpu_baseline"]["port_simd"] = cpu
    else:
        if ref_here is not None:
            out["reference_numpy_this_host"] = ref_here     # the error, so that its absence is explained on the line
        if cpu is not None:
            out["cpu_baseline"] = cpu
    if _CPU_BUDGET:
        out["host_cpu_budget"] = dict(_CPU_BUDGET, note="logical CPUs this container sees vs the CPUs' worth of run time it gets (one busy "
                                      "loop per logical CPU for 1 s, CPU time / wall time): the CPU legs use the latter as their worker / thread count")
    ref_path = os.path.join(ROOT, "profiles", "reference_cpu_timing.json")
    if os.path.exists(ref_path):      # the same measurement made in the BUILD container (8 cores), for comparison
        with open(ref_path) as f:
            rt = json.load(f)
        out["reference_numpy_build_container"] = {"dcf_per_s_one_core": rt.get("single", rt).get("dcf_per_s_one_core", rt.get("dcf_per_s_per_core")),
                                                  "dcf_per_s_pool": rt.get("pool", {}).get("dcf_per_s_pool", rt.get("pool", {}).get("dcf_per_s_all_cores_solve_only")),
                                                  "pool_procs": rt.get("pool", {}).get("procs", rt.get("pool", {}).get("cores")),
                                                  "where": "BUILD CONTAINER (8 cores), oracle/time_reference.py --write"}
    if comm is not None:
        comm.close()
    if ctx_comm is not None:
        ctx_comm.close()
    if rank == 0:
        emit(out)


if __name__ == "__main__":
    main()
