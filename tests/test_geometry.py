"""Geometry -> strip tables + statics (raftx_build_designs).

CPU part (-m "not gpu"): the C oracle (oracle/raftx_geom_oracle.h) and the host descriptor parser
(raft_amd/geometry.py) against goldens of the LIVE reference (tests/golden/geom_units.npz: strip tables packed
from the reference's Member objects, MacCamy-Fuchs Cm tables, A_hydro_morison, C_hydro, W_hydro, V, AWP, rCB).
GPU part (-m gpu): the HIP kernels against the same goldens and against the oracle, and a full solveDynamics
through device-generated designs against the host-packed upload path."""
import json

import numpy as np
import pytest

from raft_amd import geometry as G
from raft_amd._abi import RaftxError
from raft_amd import snapshot as standin
from tests.util import rel_err, group_rel_err

FX = standin.load_fixture("geom_units.npz")
UNITS = {u["name"]: u for u in FX["units"]}
NAMES = list(UNITS)
TOL = 1e-12        # fp64 restatements: expect ~1e-15 (oracle) / ~1e-14 (device libm)


def tables_of(u):
    return G.describe_unit(json.loads(u["design_json"]), heading_adjust=float(u["heading_adjust"]))


def build(ctx, units, add_mask=0, mats=None):
    D = G.concat_units([tables_of(u) for u in units])
    nD = len(units)
    nw = len(units[0]["w"])
    Z = np.zeros((nD, 6, 6))
    M0, B0, C0 = mats if mats is not None else (Z, Z, Z)
    pose = np.array([u["pose"] for u in units])
    off = ctx.build_designs(D.member_off, D.members, D.station_off, D.stations, M0, B0, C0, nw, pose=pose,
                            rho=units[0]["rho"], g=units[0]["g"], k=units[0]["k"], add_mask=add_mask,
                            cap_off=D.cap_off, caps=D.caps)
    return off


def check_unit(ctx, u, tol):
    off = build(ctx, [u])
    gold = np.asarray(u["strips"])
    assert off[-1] == len(gold), (off[-1], len(gold))
    strips, cm = ctx.fetch_strips(off[-1], len(u["cm"]))
    # positions / arms / triads / scalars, group-wise relative to the group's largest golden magnitude
    for c0, c1 in [(0, 3), (3, 6), (6, 15), (15, 18), (18, 19), (19, 23), (23, 26)]:
        assert rel_err(strips[:, c0:c1], gold[:, c0:c1]) < tol, (u["name"], c0)
    assert np.array_equal(strips[:, 26:28], gold[:, 26:28])            # member / strip indices
    if len(u["cm"]):
        assert rel_err(cm, u["cm"]) < max(tol, 1e-11), u["name"]     # Hankel functions: libm vs scipy
    S = ctx.fetch_statics()
    assert rel_err(S["A_morison"][0], u["A_hydro_morison"]) < tol
    assert rel_err(S["C_hydro"][0], u["C_hydro"]) < tol
    assert rel_err(S["W_hydro"][0], u["W_hydro"]) < tol
    assert abs(S["props"][0, G.SP_V] / u["V"] - 1) < tol
    assert abs(S["props"][0, G.SP_AWP] / u["AWP"] - 1) < tol
    assert rel_err(S["props"][0, G.SP_RCB:G.SP_RCB + 3], u["rCB"]) < tol
    # mass / inertia / weight of the members alone (live reference with a massless RNA and no point inertias).
    # M_struc carries FrustumMOI's ill-conditioned tapered branch for caps whose hole is "tapered" by one rounding
    # error (OC4semi heave-plate bulkheads): agreement there hinges on the last bit of r**5, hence the looser gate.
    assert rel_err(S["M_struc"][0], u["M_struc_bare"]) < max(tol, 2e-9), u["name"]
    assert rel_err(S["C_struc"][0], u["C_struc_bare"]) < tol
    assert rel_err(S["W_struc"][0], u["W_struc_bare"]) < tol
    assert abs(S["props"][0, G.SP_MASS] / u["m_bare"] - 1) < tol
    assert rel_err(S["props"][0, G.SP_RCG:G.SP_RCG + 3], u["rCG_bare"]) < tol


# ------------------------------------------------------------------ CPU: parser + oracle pinned on the live reference
def test_descriptor_parser_matches_reference_member_counts():
    for u in FX["units"]:
        t = tables_of(u)
        assert t.n == len(u["member_ns"])                      # one descriptor per entry of FOWT.memberList
        assert t.station_off[-1] == len(t.stations)


def test_descriptor_broadcasting_rules():
    mi = dict(name="m", type="rigid", rA=[0, 0, -10], rB=[0, 0, 5], shape="rect", stations=[0, 1], d=[3.0, 2.0],
              t=0.05, Cd=[0.6, 0.8], Ca=[[1.0, 0.9], [0.8, 0.7]], heading=[0, 90], gamma=10.0)
    gm, gs, gc = G.describe_member(dict(mi, cap_stations=[0, 1], cap_t=[0.1, 0.2], cap_d_in=[[0, 0], [1, 0.5]]), heading=90.0)
    assert np.allclose(gc, [[0, 0.1, 0, 0], [15, 0.2, 1, 0.5]])
    assert gm[G.GM_SHAPE] == 0.0 and gm[G.GM_GAMMA] == 100.0            # vertical member: heading becomes twist
    assert np.allclose(gs[:, G.GS_D:G.GS_D + 2], [[3, 2], [3, 2]])        # side pair tiled over the stations
    assert np.allclose(gs[:, G.GS_CD + 1], 0.6) and np.allclose(gs[:, G.GS_CD + 2], 0.8)   # 1-D list = [p1, p2]
    assert np.allclose(gs[:, G.GS_CA + 1], [1.0, 0.8]) and np.allclose(gs[:, G.GS_CA + 2], [0.9, 0.7])
    assert np.allclose(gs[:, G.GS_S], [0, 15])
    with pytest.raises(ValueError):
        G.describe_member(dict(mi, rA=[0, 0, 0]))
    with pytest.raises(ValueError):
        G.describe_member(dict(mi, stations=[1, 0]))
    with pytest.raises(G.UnsupportedMember):
        G.describe_member(dict(mi, type="beam"))


@pytest.mark.parametrize("name", NAMES)
def test_oracle_geometry_against_live_reference(name, oracle_ctx):
    check_unit(oracle_ctx, UNITS[name], TOL)


def test_oracle_generated_designs_solve_like_uploaded_ones(oracle_lib):
    """build_designs must leave the ctx in the same state as upload_designs of the host-packed table."""
    u = UNITS["VolturnUS-S-test@pose"]
    nw = len(u["w"])
    rng = np.random.default_rng(5)
    M0 = (np.asarray(u["M_struc"]) + np.asarray(u["A_hydro_morison"]))[None]
    B0 = np.zeros((1, 6, 6))
    C0 = (np.diag([7e4, 7e4, 0, 0, 0, 1e8]) + np.asarray(u["C_hydro"]) + np.asarray(u["C_struc"]))[None]
    zeta = rng.uniform(0.05, 0.4, size=(1, 1, nw))
    beta = np.array([[0.3]])
    out = []
    for route in ("generated", "uploaded"):
        ctx = oracle_lib.context(0)
        if route == "generated":
            Ms = M0 - np.asarray(u["A_hydro_morison"])[None] - np.asarray(u["M_struc_bare"])[None]
            Cs = C0 - np.asarray(u["C_hydro"])[None] - np.asarray(u["C_struc_bare"])[None]
            build(ctx, [u], add_mask=G.ADD_MORISON | G.ADD_HYDROSTATIC | G.ADD_INERTIA, mats=(Ms, B0, Cs))
        else:
            off = np.array([0, len(u["strips"])], dtype=np.int64)
            ctx.upload_designs_raw(off, u["strips"], M0, B0, C0, nw, None, np.array([0, len(u["cm"])]), u["cm"])
        ctx.upload_cases(u["w"], u["k"], 200.0, u["rho"], u["g"], zeta, beta)
        out.append(ctx.solve_dynamics(6, 0.01, 0.1))
        ctx.close()
    assert np.array_equal(out[0]["niter"], out[1]["niter"])
    assert group_rel_err(out[0]["Xi"][0, 0], out[1]["Xi"][0, 0]) < 1e-10


def test_build_designs_argument_errors(oracle_ctx):
    u = UNITS["OC4semi"]
    mo, mem, so, st = G.concat_units([tables_of(u)])
    Z = np.zeros((1, 6, 6))
    with pytest.raises(RaftxError):                                     # MacCamy-Fuchs member without wave numbers
        oracle_ctx.build_designs(mo, mem, so, st, Z, Z, Z, len(u["w"]), k=None)
    # a bulkhead closer to the member end than its own thickness: the reference raises ValueError (raft_member.py:684-688)
    t = tables_of(UNITS["OC3spar"])
    bad = G.MemberTable(list(t.members), [t.stations[t.station_off[i]:t.station_off[i + 1]] for i in range(t.n)],
                        [np.array([[0.05, 0.2, 0.0, 0.0]])] + [np.zeros((0, 4))] * (t.n - 1))
    D = G.concat_units([bad])
    with pytest.raises(RaftxError):
        oracle_ctx.build_designs(D.member_off, D.members, D.station_off, D.stations, Z, Z, Z, 4, cap_off=D.cap_off, caps=D.caps)


# ------------------------------------------------------------------ GPU: the HIP kernels
@pytest.mark.gpu
@pytest.mark.parametrize("name", NAMES)
def test_hip_geometry_against_live_reference(name, hip_ctx):
    check_unit(hip_ctx, UNITS[name], 1e-11)


@pytest.mark.gpu
def test_hip_geometry_batch_matches_oracle(hip_ctx, oracle_ctx):
    """All same-grid units in ONE launch (different member counts, poses, headings): offsets, records, statics."""
    nw = len(UNITS["C3-variant-0"]["w"])
    units = [u for u in FX["units"] if len(u["w"]) == nw and not len(u["cm"])] * 3
    assert len(units) >= 9
    off_h = build(hip_ctx, units)
    off_o = build(oracle_ctx, units)
    assert np.array_equal(off_h, off_o)
    sh, _ = hip_ctx.fetch_strips(off_h[-1])
    so, _ = oracle_ctx.fetch_strips(off_o[-1])
    assert rel_err(sh[:, :26], so[:, :26]) < 1e-12
    assert np.array_equal(sh[:, 26:28], so[:, 26:28])
    Sh, So = hip_ctx.fetch_statics(), oracle_ctx.fetch_statics()
    for key in ("A_morison", "C_hydro", "W_hydro", "M_struc", "C_struc", "W_struc", "props"):
        assert rel_err(Sh[key], So[key]) < 1e-12, key


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["VolturnUS-S-test@pose", "OC4semi@heel", "C3-variant-1"])
def test_hip_generated_designs_solve_like_uploaded_ones(name, hip_lib):
    u = UNITS[name]
    nw = len(u["w"])
    rng = np.random.default_rng(7)
    M0 = (np.asarray(u["M_struc"]) + np.asarray(u["A_hydro_morison"]))[None]
    B0 = np.zeros((1, 6, 6))
    C0 = (np.diag([7e4, 7e4, 0, 0, 0, 1e8]) + np.asarray(u["C_hydro"]) + np.asarray(u["C_struc"]))[None]
    zeta = rng.uniform(0.05, 0.4, size=(2, 2, nw))
    beta = np.array([[0.3, -1.0], [2.0, 0.0]])
    out = []
    for route in ("generated", "uploaded"):
        ctx = hip_lib.context(0)
        if route == "generated":
            Ms = M0 - np.asarray(u["A_hydro_morison"])[None] - np.asarray(u["M_struc_bare"])[None]
            Cs = C0 - np.asarray(u["C_hydro"])[None] - np.asarray(u["C_struc_bare"])[None]
            build(ctx, [u], add_mask=G.ADD_MORISON | G.ADD_HYDROSTATIC | G.ADD_INERTIA, mats=(Ms, B0, Cs))
        else:
            off = np.array([0, len(u["strips"])], dtype=np.int64)
            cm = u["cm"] if len(u["cm"]) else None
            ctx.upload_designs_raw(off, u["strips"], M0, B0, C0, nw, None,
                                   np.array([0, len(u["cm"])]) if cm is not None else None, cm)
        ctx.upload_cases(u["w"], u["k"], 200.0, u["rho"], u["g"], zeta, beta)
        out.append(ctx.solve_dynamics(6, 0.01, 0.1))
        ctx.close()
    assert np.array_equal(out[0]["niter"], out[1]["niter"])
    assert group_rel_err(out[0]["Xi"].reshape(-1, 6, nw), out[1]["Xi"].reshape(-1, 6, nw)) < 1e-9


# ------------------------------------------------------------------ the C3 sweep, generated instead of packed
C3 = standin.load_fixture("c3_variants.npz")


def c3_generated(ctx, n=64):
    """The first n designs of the C3 sweep (the ones the live reference built for c3_variants.npz), generated."""
    from tests.util import volturnus_sweep
    base = json.loads(FX["c3_base_json"])
    scales = np.asarray(C3["scales"])[:n]
    u0 = UNITS["C3-variant-0"]
    M_rna = np.asarray(u0["M_struc"]) - np.asarray(u0["M_struc_bare"])
    C_rest = np.asarray(u0["C_struc"]) - np.asarray(u0["C_struc_bare"]) + np.diag([7e4, 7e4, 0, 0, 0, 1e8])
    D = volturnus_sweep(base, scales).tables()
    M0 = np.repeat(M_rna[None], n, axis=0)
    C0 = np.repeat(C_rest[None], n, axis=0)
    off = ctx.build_designs(D.member_off, D.members, D.station_off, D.stations, M0, np.asarray(C3["B0"])[:n], C0,
                            len(C3["w"]), cap_off=D.cap_off, caps=D.caps,
                            add_mask=G.ADD_MORISON | G.ADD_HYDROSTATIC | G.ADD_INERTIA)
    return off, M0, C0


def test_vectorised_sweep_descriptors_equal_parsed_ones():
    from tests.util import volturnus_sweep
    sw = volturnus_sweep(json.loads(FX["c3_base_json"]), np.asarray(FX["c3_scales"]))
    for i in range(3):
        t = tables_of(UNITS["C3-variant-%d" % i])
        assert np.array_equal(sw.members[i], t.members)
        assert np.array_equal(sw.stations[i], t.stations)
        assert np.array_equal(sw.caps[i], t.caps)
    D = sw.tables()
    assert D.n_design == 3 and D.station_off[-1] == len(D.stations) and D.cap_off[-1] == len(D.caps)


def check_c3(ctx, tol):
    n = 64
    off, M_extra, C_extra = c3_generated(ctx, n)
    assert np.array_equal(off, np.asarray(C3["strip_offsets"]))
    strips, _ = ctx.fetch_strips(off[-1])
    assert rel_err(strips[:, :26], np.asarray(C3["strips"])[:, :26]) < tol
    S = ctx.fetch_statics()
    assert rel_err(S["M_struc"] + S["A_morison"] + M_extra, C3["M0"]) < tol
    assert rel_err(S["C_struc"] + S["C_hydro"] + C_extra, C3["C0"]) < tol


def test_oracle_generates_the_reference_built_c3_variants(oracle_ctx):
    check_c3(oracle_ctx, TOL)


@pytest.mark.gpu
def test_hip_generates_the_reference_built_c3_variants(hip_ctx):
    check_c3(hip_ctx, 1e-11)


@pytest.mark.gpu
def test_hip_c3_from_member_descriptions_to_reference_responses(hip_ctx):
    """Whole device pipeline: member descriptions -> strips + statics -> solveDynamics, against the LIVE reference's
    solveDynamics of the same variants (group-relative 1e-9, identical iteration counts)."""
    c3_generated(hip_ctx, 64)
    hip_ctx.upload_cases(C3["w"], C3["k"], float(C3["depth"]), 1025.0, 9.81, np.asarray(C3["zeta"])[None],
                         np.asarray(C3["beta"])[None])
    out = hip_ctx.solve_dynamics(int(C3["nIter"]), 0.01, float(C3["XiStart"]))
    assert len(C3["solved"]) >= 4
    for j, sol in enumerate(C3["solved"]):
        assert int(out["niter"][j, 0]) == int(sol["units"][0]["niter"])
        assert group_rel_err(out["Xi"][j, 0, :1], np.asarray(sol["Xi"])[:1]) < 1e-9


# ------------------------------------------------------------------ one-call sweep crossing (raftx_sweep_stats)
def _c3_crossing_inputs(n):
    from tests.util import volturnus_sweep
    base = json.loads(FX["c3_base_json"])
    scales = np.random.default_rng(0).uniform(0.75, 1.25, size=(n, 5))
    u0 = UNITS["C3-variant-0"]
    M_rna = np.asarray(u0["M_struc"]) - np.asarray(u0["M_struc_bare"])
    C_rest = np.asarray(u0["C_struc"]) - np.asarray(u0["C_struc_bare"]) + np.diag([7e4, 7e4, 0, 0, 0, 1e8])
    D = volturnus_sweep(base, scales).tables()
    return D, np.repeat(M_rna[None], n, axis=0), np.repeat(np.asarray(C3["B0"])[:1], n, axis=0), np.repeat(C_rest[None], n, axis=0)


def check_crossing(ctx, n, n_chunk, n_worker):
    """raftx_sweep_stats == build_designs + upload_cases + solve + motion_stats + fetch_results, bit for bit, whatever
    the chunking; the first 64 designs against the live reference's solveDynamics."""
    D, M0, B0, C0 = _c3_crossing_inputs(n)
    zeta2 = np.stack([np.asarray(C3["zeta"]), 0.5 * np.asarray(C3["zeta"])])          # two sea states
    beta2 = np.stack([np.asarray(C3["beta"]), np.asarray(C3["beta"]) + 0.4])
    nw = len(C3["w"])
    got = ctx.sweep_stats(D, M0, B0, C0, C3["w"], C3["k"], float(C3["depth"]), zeta2, beta2, int(C3["nIter"]), 0.01,
                          float(C3["XiStart"]), n_chunk=n_chunk, n_worker=n_worker, want_Xi=True)
    off = ctx.build_designs(D.member_off, D.members, D.station_off, D.stations, M0, B0, C0, nw, cap_off=D.cap_off,
                            caps=D.caps, add_mask=7)
    ctx.upload_cases(C3["w"], C3["k"], float(C3["depth"]), 1025.0, 9.81, zeta2, beta2)
    ctx.solve_dynamics_device(int(C3["nIter"]), 0.01, float(C3["XiStart"]))
    std, _ = ctx.motion_stats(float(C3["w"][1] - C3["w"][0]))
    ref = ctx.fetch_results(want_Xi=True)
    assert np.array_equal(got["strip_off"], off)
    assert np.array_equal(got["Xi"].view(np.uint64), ref["Xi"].view(np.uint64))
    assert np.array_equal(got["std"].view(np.uint64), std.view(np.uint64))
    assert np.array_equal(got["niter"], ref["niter"]) and np.array_equal(got["flags"], ref["flags"])
    for j, sol in enumerate(C3["solved"][:min(n, 64)]):
        assert int(got["niter"][j, 0]) == int(sol["units"][0]["niter"])
        assert group_rel_err(got["Xi"][j, 0, :1], np.asarray(sol["Xi"])[:1]) < 1e-9
    return got


def test_oracle_sweep_crossing_is_the_plain_sequence(oracle_ctx):
    check_crossing(oracle_ctx, 6, 0, 0)


@pytest.mark.gpu
@pytest.mark.parametrize("n,n_chunk,n_worker", [(70, 0, 0), (70, 1, 1), (333, 7, 3), (5, 9, 4), (1, 0, 0)])
def test_hip_sweep_crossing_chunked_equals_the_plain_sequence(hip_ctx, n, n_chunk, n_worker):
    got = check_crossing(hip_ctx, n, n_chunk, n_worker)
    assert got["timing_ms"][0] > 0 and got["timing_ms"][2] > 0


@pytest.mark.gpu
@pytest.mark.parametrize("slab_pairs,streams", [(100, 1), (64, 3), (1, 2)])
def test_hip_sweep_crossing_with_responses_in_slabs(slab_pairs, streams):
    """A crossing that downloads its responses cuts its fused launch into slabs of the pair list, each with its own
    download (raftx_hip.hip SlabPlan; one residency round per slab by default -- more pairs than these batches have).
    Forced down to 100 / 64 / 1 pairs per slab, on one to three slab streams, in a process of its own (the setting is read
    once): still the plain sequence bit for bit, whole batch and chunked."""
    import os
    import subprocess
    import sys
    env = dict(os.environ, RAFTX_XI_SLAB_PAIRS=str(slab_pairs), RAFTX_XI_SLAB_STREAMS=str(streams))
    code = ("import tests.test_geometry as t; from raft_amd import backend; ctx = backend.hip_library().context(0); "
            "t.check_crossing(ctx, %d, 0, 0); t.check_crossing(ctx, %d, 3, 2); t.check_crossing(ctx, 7, 0, 0); print('slabs ok')"
            % ((333, 200) if slab_pairs > 1 else (40, 25)))
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600,
                       cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    assert r.returncode == 0 and "slabs ok" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


@pytest.mark.gpu
def test_hip_sweep_crossing_reuses_its_worker_streams_and_reports_errors(hip_ctx):
    """Several crossings on one context (worker contexts and their memory pools are kept), then a bad batch: the error of
    the failing worker comes back through raftx_last_error, and the context still works afterwards."""
    from raft_amd._abi import RaftxError
    for n in (40, 9, 40):
        check_crossing(hip_ctx, n, 4, 2)
    D, M0, B0, C0 = _c3_crossing_inputs(8)
    bad = G.DesignTables(D.member_off, D.members.copy(), D.station_off, D.stations, D.cap_off, D.caps)
    bad.members[int(D.member_off[5]), G.GM_DLSMAX] = -1.0                 # design 5: invalid strip length
    with pytest.raises(RaftxError, match="dlsMax"):
        hip_ctx.sweep_stats(bad, M0, B0, C0, C3["w"], C3["k"], float(C3["depth"]), C3["zeta"], C3["beta"], int(C3["nIter"]),
                            n_chunk=4, n_worker=2)
    check_crossing(hip_ctx, 12, 3, 3)


def check_streamed_crossings(ctx):
    """Two crossings in flight (raftx_sweep_submit on slots 0 and 1, then raftx_sweep_wait): different batches, the
    second submitted before the first is collected; each equals its blocking raftx_sweep_stats bit for bit.  Misuse is
    reported: a busy slot cannot be submitted to, an idle one cannot be waited for."""
    from raft_amd._abi import RaftxError
    nw = len(C3["w"])
    args = (C3["w"], C3["k"], float(C3["depth"]), np.asarray(C3["zeta"])[None], np.asarray(C3["beta"])[None], int(C3["nIter"]), 0.01,
            float(C3["XiStart"]))
    batches = [_c3_crossing_inputs(n) for n in (37, 70, 12)]
    want = [ctx.sweep_stats(D, M0, B0, C0, *args, want_Xi=True) for D, M0, B0, C0 in batches]
    h = [None, None]
    got = []
    h[0] = ctx.sweep_submit(0, *batches[0], *args, want_Xi=True)
    with pytest.raises(RaftxError, match="still in flight"):
        ctx.sweep_submit(0, *batches[1], *args)
    h[1] = ctx.sweep_submit(1, *batches[1], *args, want_Xi=True)
    got.append(ctx.sweep_wait(h[0]))
    h[0] = ctx.sweep_submit(0, *batches[2], *args, want_Xi=True)          # slot 0 again while slot 1 is in flight
    got.append(ctx.sweep_wait(h[1]))
    got.append(ctx.sweep_wait(h[0]))
    with pytest.raises(RaftxError, match="nothing submitted"):
        ctx.sweep_wait(h[0])
    for g_, w_ in zip(got, want):
        for key in ("Xi", "std"):
            assert np.array_equal(g_[key].view(np.uint64), w_[key].view(np.uint64)), key
        assert np.array_equal(g_["niter"], w_["niter"]) and np.array_equal(g_["flags"], w_["flags"])
        assert np.array_equal(g_["strip_off"], w_["strip_off"])
    assert nw == got[0]["Xi"].shape[-1]


def check_staged_crossings(ctx):
    """Three crossings in flight through the staged form (raftx_sweep_prepare / _launch / _wait on slots 0 .. 2, in the
    order prepare(i+2), launch(i+1), wait(i) of a long sweep): every batch equals its blocking raftx_sweep_stats bit for
    bit.  Misuse is reported: launching a slot that was not prepared, preparing one that is still in flight."""
    from raft_amd._abi import RaftxError
    args = (C3["w"], C3["k"], float(C3["depth"]), np.asarray(C3["zeta"])[None], np.asarray(C3["beta"])[None], int(C3["nIter"]), 0.01,
            float(C3["XiStart"]))
    sizes = (23, 41, 9, 30, 17)
    batches = [_c3_crossing_inputs(n) for n in sizes]
    want = [ctx.sweep_stats(D, M0, B0, C0, *args, want_Xi=True) for D, M0, B0, C0 in batches]
    with pytest.raises(RaftxError, match="nothing prepared"):
        ctx.sweep_launch(dict(slot=2))
    n = len(batches)
    sub = lambda i: ctx.sweep_prepare(i % 3, *batches[i], *args, want_Xi=True)
    hs = {0: sub(0), 1: sub(1)}
    with pytest.raises(RaftxError, match="still in flight"):
        ctx.sweep_prepare(1, *batches[1], *args)
    ctx.sweep_launch(hs[0])
    got = []
    for i in range(n):
        if i + 2 < n:
            hs[i + 2] = sub(i + 2)
        if i + 1 < n:
            ctx.sweep_launch(hs[i + 1])
        got.append(ctx.sweep_wait(hs.pop(i)))
    for g_, w_ in zip(got, want):
        for key in ("Xi", "std"):
            assert np.array_equal(g_[key].view(np.uint64), w_[key].view(np.uint64)), key
        assert np.array_equal(g_["niter"], w_["niter"]) and np.array_equal(g_["flags"], w_["flags"])
        assert np.array_equal(g_["strip_off"], w_["strip_off"])


def check_staged_crossings_with_changing_sea_states(ctx):
    """A batch is solved with the sea states it was PREPARED with: three batches in flight, each with its own sea states
    (other amplitudes, headings, number of cases), prepared before the batch ahead of them is launched -- every batch
    equals its blocking call.  A prepared batch can be cancelled (its outputs untouched) and its slot used again."""
    from raft_amd._abi import RaftxError
    z0, b0 = np.asarray(C3["zeta"]), np.asarray(C3["beta"])
    seas = [(z0[None], b0[None]),
            (np.stack([0.5 * z0, 1.5 * z0]), np.stack([b0 + 0.3, b0 - 0.7])),
            (0.8 * z0[None], b0[None] + 1.1),
            (z0[None], b0[None])]
    fixed = lambda z, b: (C3["w"], C3["k"], float(C3["depth"]), z, b, int(C3["nIter"]), 0.01, float(C3["XiStart"]))
    batches = [_c3_crossing_inputs(n) for n in (19, 33, 8, 27)]
    want = [ctx.sweep_stats(*bt, *fixed(*sea), want_Xi=True) for bt, sea in zip(batches, seas)]
    sub = lambda i: ctx.sweep_prepare(i % 3, *batches[i], *fixed(*seas[i]), want_Xi=True)
    hs = {0: sub(0), 1: sub(1)}
    ctx.sweep_launch(hs[0])
    got = []
    for i in range(4):
        if i + 2 < 4:
            hs[i + 2] = sub(i + 2)
        if i + 1 < 4:
            ctx.sweep_launch(hs[i + 1])
        got.append(ctx.sweep_wait(hs.pop(i)))
    for g_, w_ in zip(got, want):
        for key in ("Xi", "std"):
            assert g_[key].shape == w_[key].shape and np.array_equal(g_[key].view(np.uint64), w_[key].view(np.uint64)), key
        assert np.array_equal(g_["niter"], w_["niter"]) and np.array_equal(g_["flags"], w_["flags"])
    # cancel: slot 1 prepared with yet other sea states, never launched
    h = ctx.sweep_prepare(1, *batches[1], *fixed(0.3 * z0[None], b0[None] + 2.0), want_Xi=True)
    h["out"]["std"][...] = -7.0
    ctx.sweep_cancel(h)
    assert np.all(h["out"]["std"] == -7.0)
    with pytest.raises(RaftxError, match="nothing prepared"):
        ctx.sweep_launch(h)
    again = ctx.sweep_wait(ctx.sweep_submit(1, *batches[1], *fixed(*seas[1]), want_Xi=True))
    assert np.array_equal(again["Xi"].view(np.uint64), want[1]["Xi"].view(np.uint64))
    ctx.sweep_cancel(dict(slot=2))                        # idle slot: no-op


def check_soak(ctx, n_step):
    """A long streak through the four slots (prepare(i+3), launch(i+2), wait(i)): three batch sizes x two sets of sea states
    in rotation, responses downloaded -- every collected batch equals its blocking call bit for bit, pools and slots are
    recycled hundreds of times (the soak the sanitised build runs too, scripts/gpu_asan.sh)."""
    z0, b0 = np.asarray(C3["zeta"]), np.asarray(C3["beta"])
    seas = [(z0[None], b0[None]), (np.stack([0.5 * z0, 1.5 * z0]), np.stack([b0 + 0.3, b0 - 0.7]))]
    fixed = lambda z, b: (C3["w"], C3["k"], float(C3["depth"]), z, b, int(C3["nIter"]), 0.01, float(C3["XiStart"]))
    batches = [_c3_crossing_inputs(n) for n in (21, 8, 34)]
    combos = [(bi, si) for bi in range(3) for si in range(2)]
    want = {c: ctx.sweep_stats(*batches[c[0]], *fixed(*seas[c[1]]), want_Xi=True) for c in combos}
    pick = lambda i: combos[(i * 5 + i // 7) % len(combos)]
    sub = lambda i: ctx.sweep_prepare(i % 4, *batches[pick(i)[0]], *fixed(*seas[pick(i)[1]]), want_Xi=True)
    hs = {i: sub(i) for i in range(min(n_step, 3))}
    for i in range(min(n_step, 2)):
        ctx.sweep_launch(hs[i])
    for i in range(n_step):
        if i + 3 < n_step:
            hs[i + 3] = sub(i + 3)
        if i + 2 < n_step:
            ctx.sweep_launch(hs[i + 2])
        g_, w_ = ctx.sweep_wait(hs.pop(i)), want[pick(i)]
        assert np.array_equal(g_["Xi"].view(np.uint64), w_["Xi"].view(np.uint64)), i
        assert np.array_equal(g_["std"].view(np.uint64), w_["std"].view(np.uint64)) and np.array_equal(g_["niter"], w_["niter"]), i


def test_oracle_streamed_crossings(oracle_ctx):
    check_streamed_crossings(oracle_ctx)
    check_staged_crossings(oracle_ctx)
    check_staged_crossings_with_changing_sea_states(oracle_ctx)
    check_soak(oracle_ctx, 9)


@pytest.mark.gpu
def test_hip_streamed_crossings(hip_ctx):
    check_streamed_crossings(hip_ctx)
    check_staged_crossings(hip_ctx)
    check_staged_crossings_with_changing_sea_states(hip_ctx)
    check_crossing(hip_ctx, 20, 0, 0)                     # the blocking call still works on the same context afterwards


@pytest.mark.gpu
def test_hip_fused_generation_is_the_generation_kernel_bit_for_bit(hip_ctx):
    """Streamed crossings with one sea state per design build their strip tables INSIDE the fused fixed point
    (raftx_kpg_f0, raft_amd/csrc/raftx_fusedgen.h: the workgroup that has claimed a design generates its tables, then solves
    it) when RAFTX_FUSED_GEN=1; the default keeps k_geom_design + k_geom_addup as kernels of their own (faster, measured).  Same batches both ways, three in
    flight: responses, statistics, iteration counts and strip offsets agree bit for bit, and raftx_sweep_generation says
    which form ran.  Two sea states per design (a design claimed twice) never take the fused form."""
    import os
    ctx = hip_ctx
    z0, b0 = np.asarray(C3["zeta"]), np.asarray(C3["beta"])
    fixed = lambda z, b: (C3["w"], C3["k"], float(C3["depth"]), z, b, int(C3["nIter"]), 0.01, float(C3["XiStart"]))
    sizes = (61, 300, 7, 1, 129)
    batches = [_c3_crossing_inputs(n) for n in sizes]

    def stream(sea, want_Xi):
        hs, got = {}, []
        sub = lambda i: ctx.sweep_prepare(i % 3, *batches[i], *fixed(*sea), want_Xi=want_Xi)
        hs[0], hs[1] = sub(0), sub(1)
        ctx.sweep_launch(hs[0])
        for i in range(len(batches)):
            if i + 2 < len(batches):
                hs[i + 2] = sub(i + 2)
            if i + 1 < len(batches):
                ctx.sweep_launch(hs[i + 1])
            got.append(ctx.sweep_wait(hs.pop(i)))
        return got

    prev = os.environ.get("RAFTX_FUSED_GEN")
    try:
        runs = {}
        for mode in ("1", "0"):
            os.environ["RAFTX_FUSED_GEN"] = mode
            for want_Xi in (False, True):
                runs[mode, want_Xi] = stream((z0[None], b0[None]), want_Xi)
        os.environ["RAFTX_FUSED_GEN"] = "1"
        two = stream((np.stack([0.5 * z0, 1.5 * z0]), np.stack([b0 + 0.3, b0 - 0.7])), False)
    finally:
        if prev is None:
            os.environ.pop("RAFTX_FUSED_GEN", None)
        else:
            os.environ["RAFTX_FUSED_GEN"] = prev
    for want_Xi in (False, True):
        for i, (a, b) in enumerate(zip(runs["1", want_Xi], runs["0", want_Xi])):
            assert a["generation_fused_blocks"][1] >= 1
            assert a["generation_fused_blocks"][0] == a["generation_fused_blocks"][1], (want_Xi, i, a["generation_fused_blocks"])
            assert b["generation_fused_blocks"][0] == 0
            for key in ("std",) + (("Xi",) if want_Xi else ()):
                assert np.array_equal(a[key].view(np.uint64), b[key].view(np.uint64)), (key, i)
            assert np.array_equal(a["niter"], b["niter"]) and np.array_equal(a["flags"], b["flags"])
            assert np.array_equal(a["strip_off"], b["strip_off"])
            assert int(a["niter"].min()) >= 1 and not np.isnan(a["std"]).any()
    assert all(t["generation_fused_blocks"][0] == 0 for t in two)


@pytest.mark.gpu
def test_hip_soak_of_300_staged_crossings(hip_ctx):
    check_soak(hip_ctx, 300)


# ------------------------------------------------------------------ ballast trim (Model.adjustBallastDensity)
TRIM_NAMES = [n for n in NAMES if "trim_drho" in UNITS[n]]


def check_trim(ctx, u, tol):
    """RAFTX_TRIM_BALLAST against the live reference's adjustBallastDensity on the full model: the density change,
    the total ballast volume, and the statics after the trim (M_extra carries the rotor-nacelle assembly)."""
    D = G.concat_units([tables_of(u)])
    M_rna = (np.asarray(u["M_struc"]) - np.asarray(u["M_struc_bare"]))[None]
    C_rna = (np.asarray(u["C_struc"]) - np.asarray(u["C_struc_bare"]))[None]
    W_rna = np.asarray(u["W_struc"]) - np.asarray(u["W_struc_bare"])
    Z = np.zeros((1, 6, 6))
    ctx.build_designs(D.member_off, D.members, D.station_off, D.stations, M_rna, Z, C_rna, len(u["w"]),
                      pose=np.array(u["pose"])[None], rho=u["rho"], g=u["g"], k=u["k"], cap_off=D.cap_off, caps=D.caps,
                      add_mask=G.TRIM_BALLAST, Fz_moor=np.array([u["trim_Fz"]]))
    S = ctx.fetch_statics()
    assert abs(S["props"][0, G.SP_VFILL] / u["trim_vfill"] - 1) < tol
    assert abs(S["props"][0, G.SP_DRHO] - u["trim_drho"]) < 1e-9 * max(1.0, abs(u["trim_drho"])) + 1e-7
    assert rel_err(S["M_struc"][0] + M_rna[0], u["trim_M_struc"]) < max(tol, 2e-9)
    assert rel_err(S["C_struc"][0] + C_rna[0], u["trim_C_struc"]) < max(tol, 1e-10)
    assert rel_err(S["W_struc"][0] + W_rna, u["trim_W_struc"]) < max(tol, 1e-10)
    m_rna = M_rna[0, 0, 0]
    assert abs((S["props"][0, G.SP_MASS] + m_rna) / u["trim_m"] - 1) < max(tol, 1e-11)
    # heave balance after the trim: weight = buoyancy + mooring
    assert abs(-(S["props"][0, G.SP_MASS] + m_rna) * u["g"] + S["props"][0, G.SP_V] * u["rho"] * u["g"] + u["trim_Fz"]) \
        < 1e-9 * u["trim_m"] * u["g"]


@pytest.mark.parametrize("name", TRIM_NAMES)
def test_oracle_ballast_trim_against_live_reference(name, oracle_ctx):
    check_trim(oracle_ctx, UNITS[name], TOL)


def test_ballast_trim_needs_ballast(oracle_ctx):
    u = UNITS["OC3spar"]
    t = tables_of(u)
    st = t.stations.copy()
    st[:, G.GS_LFILL] = 0.0
    D = G.concat_units([G.MemberTable(list(t.members), [st[t.station_off[i]:t.station_off[i + 1]] for i in range(t.n)],
                                      [t.caps[t.cap_off[i]:t.cap_off[i + 1]] for i in range(t.n)])])
    Z = np.zeros((1, 6, 6))
    with pytest.raises(RaftxError):                   # the reference raises too (raft_model.py:1801-1802)
        oracle_ctx.build_designs(D.member_off, D.members, D.station_off, D.stations, Z, Z, Z, 4, cap_off=D.cap_off,
                                 caps=D.caps, add_mask=G.TRIM_BALLAST)


@pytest.mark.gpu
@pytest.mark.parametrize("name", TRIM_NAMES)
def test_hip_ballast_trim_against_live_reference(name, hip_ctx):
    check_trim(hip_ctx, UNITS[name], 1e-11)


# ------------------------------------------------------------------ empty / ragged inputs
def check_ragged(ctx):
    """No designs at all; a design whose only member is dry (no strips, no buoyancy); that design in the middle of a
    batch, solved: its response to waves is exactly zero and its neighbours are unaffected."""
    Z0 = np.zeros((0, 6, 6))
    off = ctx.build_designs(np.zeros(1, dtype=np.int64), np.zeros((0, 16)), np.zeros(1, dtype=np.int64), np.zeros((0, 16)),
                            Z0, Z0, Z0, 8)
    assert off.tolist() == [0]
    u = UNITS["OC3spar"]
    t = tables_of(u)
    tower = G.MemberTable([t.members[1]], [t.stations[t.station_off[1]:t.station_off[2]]], [t.caps[t.cap_off[1]:t.cap_off[2]]])
    assert tower.members[0, G.GM_RA + 2] > 0                        # the tower starts above the waterline
    D = G.concat_units([t, tower, t])
    nw = len(u["w"])
    M0 = np.repeat((np.eye(6) * [8e6, 8e6, 8e6, 7e9, 7e9, 2e8])[None], 3, 0)
    C0 = np.repeat(np.diag([4e4, 4e4, 3e5, 1e9, 1e9, 1e8])[None], 3, 0)
    off = ctx.build_designs(D.member_off, D.members, D.station_off, D.stations, M0, np.zeros((3, 6, 6)), C0, nw,
                            rho=u["rho"], g=u["g"], cap_off=D.cap_off, caps=D.caps, add_mask=G.ADD_MORISON)
    assert off[1] == off[2] and off[1] - off[0] == off[3] - off[2] == len(u["strips"])
    S = ctx.fetch_statics()
    assert not np.any(S["A_morison"][1]) and not np.any(S["C_hydro"][1]) and S["props"][1, G.SP_V] == 0.0
    assert S["props"][1, G.SP_MASS] > 1e5                            # ... but it has mass
    rng = np.random.default_rng(2)
    zeta = rng.uniform(0.1, 0.5, size=(1, 1, nw))
    ctx.upload_cases(u["w"], u["k"], 320.0, u["rho"], u["g"], zeta, np.array([[0.4]]))
    out = ctx.solve_dynamics(5, 0.01, 0.1)
    assert not np.any(out["Xi"][1]) and not np.any(out["flags"] & 2)
    assert np.array_equal(out["Xi"][0].view(np.uint64), out["Xi"][2].view(np.uint64)) and np.any(out["Xi"][0])


def test_oracle_empty_and_ragged_batches(oracle_ctx):
    check_ragged(oracle_ctx)


@pytest.mark.gpu
def test_hip_empty_and_ragged_batches(hip_ctx):
    check_ragged(hip_ctx)


# ------------------------------------------------------------------ the reference's OWN member-level known answers
REFMEM = standin.load_fixture("refgold_members.npz")["cases"]


def check_ref_member(ctx, case, tol):
    """tests/test_member.py:604-623 of the reference (test_inertia, test_hydrostatics, test_hydroConstants; reference
    point at the origin, rho 1025, g 9.81) through raftx_build_designs with the member as a one-member unit."""
    mi = json.loads(case["member_json"])
    gm, gs, gc = G.describe_member(mi, heading=float(np.atleast_1d(mi.get("heading", 0.0))[0]))
    D = G.concat_units([G.MemberTable([gm], [gs], [gc])])
    Z = np.zeros((1, 6, 6))
    off = ctx.build_designs(D.member_off, D.members, D.station_off, D.stations, Z, Z, Z, 4, rho=1025.0, g=9.81,
                            cap_off=D.cap_off, caps=D.caps)
    S = ctx.fetch_statics()
    strips, _ = ctx.fetch_strips(off[-1])
    sym = lambda A: 0.5 * (A + A.T)
    # test_inertia: mass matrix about the origin, total mass, centre of mass
    np.testing.assert_allclose(S["M_struc"][0], sym(case["inertiaMatrix"]), rtol=tol, atol=1e-6 * np.abs(case["inertiaMatrix"]).max())
    mshell, mfill, cgx, cgy, cgz = case["inertiaBasic"]
    assert abs(S["props"][0, G.SP_MASS] / (mshell + mfill) - 1) < tol
    np.testing.assert_allclose(S["props"][0, G.SP_RCG:G.SP_RCG + 3], [cgx, cgy, cgz], rtol=tol, atol=2e-5)
    # test_hydrostatics: buoyancy vector and stiffness about the origin (ours is the symmetrised unit-level matrix)
    np.testing.assert_allclose(S["W_hydro"][0], case["Fvec"], rtol=tol, atol=1e-5 * max(1.0, np.abs(case["Fvec"]).max()))
    # Inclined surface-piercing members: the reference's member-level matrix about the origin and its unit-level matrix
    # (member matrix about the member node, then T^T C T; raft_fowt.py:1122) differ in the waterplane block, because the
    # heave stiffness carries 1/cos(phi) and the moment terms do not (raft_member.py:931-946).  raftx reproduces the
    # unit-level one (pinned on live FOWT.C_hydro, test_*_geometry_against_live_reference), so that block is left out here.
    Cg = sym(case["Cmat"])
    q = np.asarray(mi["rB"], float) - np.asarray(mi["rA"], float)
    keep = np.ones((6, 6), dtype=bool)
    if (q[0] != 0 or q[1] != 0) and mi["rA"][2] * mi["rB"][2] < 0:
        keep[2:5, 2:5] = False
    np.testing.assert_allclose(S["C_hydro"][0][keep], Cg[keep], rtol=tol, atol=1e-6 * np.abs(case["Cmat"]).max())
    np.testing.assert_allclose(S["props"][0, G.SP_RCB:G.SP_RCB + 3], case["r_center"], rtol=tol, atol=2e-5)
    # test_hydroConstants: Morison added mass, and the inertial-excitation matrix rebuilt from the strip records
    np.testing.assert_allclose(S["A_morison"][0], case["Ahydro"], rtol=tol, atol=1e-6 * np.abs(case["Ahydro"]).max())
    I6 = np.zeros((6, 6))
    for rec in strips:
        a = rec[3:6]
        for c, n in ((rec[16], rec[9:12]), (rec[17], rec[12:15]), (rec[15], rec[6:9])):       # Ip1 p1, Ip2 p2, Iq q
            g6 = np.concatenate([n, np.cross(a, n)])
            I6 += c * np.outer(g6, g6)
    np.testing.assert_allclose(I6, case["Ihydro"], rtol=tol, atol=1e-6 * np.abs(case["Ihydro"]).max())


@pytest.mark.parametrize("i", range(len(REFMEM)))
def test_oracle_against_reference_member_known_answers(i, oracle_ctx):
    check_ref_member(oracle_ctx, REFMEM[i], 2e-5)


@pytest.mark.gpu
@pytest.mark.parametrize("i", range(len(REFMEM)))
def test_hip_against_reference_member_known_answers(i, hip_ctx):
    check_ref_member(hip_ctx, REFMEM[i], 2e-5)


# ------------------------------------------------------------------ the reference's OWN unit-level statics goldens
REFSTAT = standin.load_fixture("refgold_statics.npz")["cases"]


def check_ref_statics(ctx, c):
    """tests/test_fowt.py:63-108 of the reference (test_statics, test_hydroConstants; same rtol 1e-5 / atol 1e-3)."""
    t = G.describe_unit(json.loads(c["design_json"]))
    D = G.concat_units([t])
    Z = np.zeros((1, 6, 6))
    ctx.build_designs(D.member_off, D.members, D.station_off, D.stations, Z, Z, Z, len(c["k"]), rho=c["rho"], g=c["g"],
                      k=c["k"], cap_off=D.cap_off, caps=D.caps)
    S = ctx.fetch_statics()
    close = lambda a, b: np.testing.assert_allclose(a, b, rtol=1e-5, atol=1e-3)
    close(S["A_morison"][0], c["A_hydro_morison"])
    close(S["C_hydro"][0], c["true_C_hydro"])
    close(S["W_hydro"][0], c["true_W_hydro"])
    close(S["props"][0, G.SP_RCB:G.SP_RCB + 3], c["true_rCB"])
    close(S["M_struc"][0] + c["M_rest"], c["true_M_struc"])
    close(S["C_struc"][0] + c["C_rest"], c["true_C_struc"])
    close(S["W_struc"][0] + c["W_rest"], c["true_W_struc"])


@pytest.mark.parametrize("i", range(len(REFSTAT)))
def test_oracle_against_reference_statics_pickles(i, oracle_ctx):
    check_ref_statics(oracle_ctx, REFSTAT[i])


@pytest.mark.gpu
@pytest.mark.parametrize("i", range(len(REFSTAT)))
def test_hip_against_reference_statics_pickles(i, hip_ctx):
    check_ref_statics(hip_ctx, REFSTAT[i])


# ------------------------------------------------------------------ member description -> reference response, MCF + pose
def check_pose_mcf_end_to_end(ctx, tol):
    """VolturnUS-S test deck (MacCamy-Fuchs columns) at an offset, heeled pose: member descriptions in, device-generated
    strips + complex Cm tables + statics, solveDynamics for one and two wave headings -- against the LIVE reference's
    solveDynamics of that deck (tests/golden/pose_volturnus_mcf.npz)."""
    fx = standin.load_fixture("pose_volturnus_mcf.npz")
    fm = fx["model"]["fowts"][0]
    u = UNITS["VolturnUS-S-test@pose"]
    assert np.allclose(u["w"], fm["w"])
    D = G.concat_units([tables_of(u)])
    M_extra = (np.asarray(u["M_struc"]) - np.asarray(u["M_struc_bare"]))[None]
    C_extra = (np.asarray(u["C_struc"]) - np.asarray(u["C_struc_bare"]) + np.asarray(fm["C_moor"]) + np.asarray(fm["C_elast"]))[None]
    B0 = (np.asarray(fm["B_struc"]) + np.sum(np.asarray(fm["B_gyro"]), axis=2))[None]
    nw = len(u["w"])
    for c in fx["cases"]:
        un = c["units"][0]
        ctx.build_designs(D.member_off, D.members, D.station_off, D.stations, M_extra, B0, C_extra, nw,
                          pose=np.array(u["pose"])[None], rho=u["rho"], g=u["g"], k=u["k"], cap_off=D.cap_off, caps=D.caps,
                          add_mask=G.ADD_MORISON | G.ADD_HYDROSTATIC | G.ADD_INERTIA)
        ctx.upload_cases(u["w"], u["k"], float(fm["depth"]), u["rho"], u["g"], np.asarray(un["zeta"])[None],
                         np.asarray(un["beta"])[None])
        out = ctx.solve_dynamics(int(fx["model"]["nIter"]), 0.01, float(fx["model"]["XiStart"]))
        nH = len(un["beta"])
        assert int(out["niter"][0, 0]) == int(un["niter"])
        assert group_rel_err(out["Xi"][0, 0], np.asarray(c["Xi"])[:nH]) < tol


def test_oracle_member_descriptions_to_reference_response_mcf_pose(oracle_ctx):
    check_pose_mcf_end_to_end(oracle_ctx, 1e-9)


@pytest.mark.gpu
def test_hip_member_descriptions_to_reference_response_mcf_pose(hip_ctx):
    check_pose_mcf_end_to_end(hip_ctx, 1e-9)


def check_long_runs(ctx_a, ctx_b, dls):
    """A spar discretised finely: runs of more than 64 strips are cut at the cap (and continue as a new run), designs with
    more strips than the generation kernel has lanes take its serial run detection; generated tables and responses of
    the two libraries agree."""
    u = UNITS["OC3spar"]
    t = G.concat_units([tables_of(u)])
    mem = t.members.copy()
    mem[:, G.GM_DLSMAX] = dls
    nw = len(u["w"])
    M0 = (np.eye(6) * [8e6, 8e6, 8e6, 7e9, 7e9, 2e8])[None]
    C0 = np.diag([4e4, 4e4, 3e5, 1e9, 1e9, 1e8])[None]
    rng = np.random.default_rng(5)
    zeta = rng.uniform(0.1, 0.5, size=(1, 1, nw))
    outs, strips = [], []
    for ctx in (ctx_a, ctx_b):
        off = ctx.build_designs(t.member_off, mem, t.station_off, t.stations, M0, np.zeros((1, 6, 6)), C0, nw,
                                rho=u["rho"], g=u["g"], cap_off=t.cap_off, caps=t.caps, add_mask=G.ADD_MORISON)
        strips.append(ctx.fetch_strips(off[-1])[0])
        ctx.upload_cases(u["w"], u["k"], 320.0, u["rho"], u["g"], zeta, np.array([[0.3]]))
        outs.append(ctx.solve_dynamics(6, 0.01, 0.1))
    assert len(strips[0]) == len(strips[1]) > 64
    assert rel_err(strips[0][:, :26], strips[1][:, :26]) < TOL
    assert np.array_equal(strips[0][:, 26:28], strips[1][:, 26:28])
    assert np.array_equal(outs[0]["niter"], outs[1]["niter"])
    assert rel_err(outs[0]["Xi"], outs[1]["Xi"]) < 1e-9 and np.any(outs[0]["Xi"])
    return len(strips[0])


@pytest.mark.gpu
@pytest.mark.parametrize("dls,more_than", [(1.6, 64), (0.8, 128)])
def test_hip_runs_longer_than_the_cap(hip_ctx, oracle_ctx, dls, more_than):
    assert check_long_runs(hip_ctx, oracle_ctx, dls) > more_than


# ------------------------------------------------------------------ parametric variants expanded by the library
# (raftx_variant_program / raftx_expand_variants / raftx_sweep_prepare_variants; raft/parametersweep.py:39-87)
def _c3_variant_sweep(n, rows=0):
    from raft_amd import geometry as G
    from raft_amd.sweep import VariantSweep
    base = json.loads(FX["c3_base_json"])
    scales = np.random.default_rng(0).uniform(0.75, 1.25, size=(rows + n, 5))[rows:]
    _, M0, B0, C0 = _c3_crossing_inputs(1)
    rep = lambda a: np.repeat(a[:1], n, axis=0)
    return VariantSweep(G.volturnus_program(base), G.volturnus_params(scales), rep(M0), rep(B0), rep(C0), C3["w"], C3["k"],
                        float(C3["depth"]), np.asarray(C3["zeta"])[None], np.asarray(C3["beta"])[None], int(C3["nIter"]),
                        float(C3["XiStart"])), scales


def check_variant_expansion(ctx, n):
    """The library's expansion of the C3 program == raft_amd.geometry.volturnus_sweep (NumPy, the reference's edits
    parametersweep.py:56-87 vectorised) BIT FOR BIT: members, stations, caps and the uniform offsets."""
    from tests.util import volturnus_sweep
    vs, scales = _c3_variant_sweep(n)
    D = volturnus_sweep(json.loads(FX["c3_base_json"]), scales).tables()
    T = vs.expanded_tables(ctx)
    for name in ("members", "stations", "caps"):
        a, b = np.ascontiguousarray(getattr(T, name)), np.ascontiguousarray(getattr(D, name))
        assert a.shape == b.shape and np.array_equal(a.view(np.uint64), b.view(np.uint64)), name
    for name in ("member_off", "station_off", "cap_off"):
        assert np.array_equal(getattr(T, name), getattr(D, name)), name


def check_variant_crossing(ctx, n, n_chunk):
    """A crossing fed with PARAMETERS (descriptors written by the library) == the crossing fed with the host's descriptor
    arrays, bit for bit; streamed batches with new parameter rows per batch keep the slots configured."""
    vs, scales = _c3_variant_sweep(n)
    D, M0, B0, C0 = _c3_crossing_inputs(n)
    want = ctx.sweep_stats(D, M0, B0, C0, C3["w"], C3["k"], float(C3["depth"]), np.asarray(C3["zeta"])[None], np.asarray(C3["beta"])[None],
                           int(C3["nIter"]), 0.01, float(C3["XiStart"]), n_chunk=n_chunk, want_Xi=True)
    got = vs.run_crossing(ctx, n_chunk=n_chunk, want_Xi=True)
    assert np.array_equal(got["strip_off"], want["strip_off"])
    assert np.array_equal(got["Xi"].view(np.uint64), want["Xi"].view(np.uint64))
    assert np.array_equal(got["std"].view(np.uint64), want["std"].view(np.uint64)) and np.array_equal(got["niter"], want["niter"])
    # a stream of batches with DISTINCT candidates: submit(i + 1) before wait(i), new parameter rows each time
    from raft_amd import geometry as G
    batches = [G.volturnus_params(np.random.default_rng(0).uniform(0.75, 1.25, size=((i + 1) * n, 5))[i * n:]) for i in range(4)]
    sweeps = [_c3_variant_sweep(n, rows=i * n)[0] for i in range(4)]
    alone = [s.run_crossing(ctx, n_chunk=n_chunk) for s in sweeps]
    vs.set_params(batches[0])
    h = vs.submit_crossing(ctx, 0, n_chunk=n_chunk)
    for i in range(4):
        h_next = None
        if i + 1 < 4:
            vs.set_params(batches[i + 1])                                # a new parameter array; the handle in flight keeps its own
            h_next = vs.submit_crossing(ctx, (i + 1) % 2, n_chunk=n_chunk)
        r = vs.wait_crossing(ctx, h)
        assert np.array_equal(r["std"].view(np.uint64), alone[i]["std"].view(np.uint64)) and np.array_equal(r["niter"], alone[i]["niter"]), i
        h = h_next
    assert not np.array_equal(alone[0]["std"], alone[1]["std"])          # the batches ARE different candidates
    # a program cannot be replaced under a batch in flight
    vs.set_params(batches[1])
    h = vs.submit_crossing(ctx, 0, n_chunk=n_chunk)
    if ctx.rlib.is_device:
        with pytest.raises(RaftxError, match="still in flight"):
            ctx.variant_program(vs.program)
    vs.wait_crossing(ctx, h)
    # resident form (upload + solve) sees the same rows
    vs.set_params(batches[0])
    res = vs.run(ctx)
    assert np.array_equal(res["Xi"].reshape(want["Xi"].shape).view(np.uint64), want["Xi"].view(np.uint64))


def check_variant_edge_batches(ctx):
    """one candidate, and none at all: a batch of zero variants crosses the boundary and comes back empty"""
    check_variant_crossing(ctx, 1, 0)
    vs, _ = _c3_variant_sweep(3)
    vs0 = vs.take(0, 0)
    out = vs0.run_crossing(ctx)
    assert out["std"].shape == (0, 1, 6) and out["niter"].shape == (0, 1) and np.array_equal(out["strip_off"], [0])
    gm, gs, gc = ctx.expand_variants(np.zeros((0, 5)))
    assert gm.shape[0] == 0 and gs.shape[0] == 0 and gc.shape[0] == 0


def test_oracle_variant_program(oracle_ctx):
    check_variant_expansion(oracle_ctx, 40)
    check_variant_crossing(oracle_ctx, 5, 0)
    check_variant_edge_batches(oracle_ctx)


def test_variant_program_argument_errors(oracle_ctx):
    from raft_amd import geometry as G
    with pytest.raises(RaftxError, match="no program"):
        oracle_ctx.variant_program(None)
        oracle_ctx._vprog = (1, 1, 0, 5)
        oracle_ctx.expand_variants(np.zeros((2, 5)))
    P = G.volturnus_program(json.loads(FX["c3_base_json"]))
    oracle_ctx.variant_program(P)
    with pytest.raises(ValueError):
        oracle_ctx.expand_variants(np.zeros((2, 4)))                     # wrong number of parameters
    vs, _ = _c3_variant_sweep(3)
    with pytest.raises(ValueError, match="same size"):
        vs.set_params(np.zeros((4, 5)))


@pytest.mark.gpu
def test_hip_variant_program(hip_ctx, oracle_ctx):
    check_variant_expansion(hip_ctx, 300)
    check_variant_crossing(hip_ctx, 70, 0)
    check_variant_crossing(hip_ctx, 333, 3)
    check_variant_edge_batches(hip_ctx)
    # ... and the oracle's expansion of the same program is the same bits
    vs, _ = _c3_variant_sweep(50)
    a, b = vs.expanded_tables(hip_ctx), _c3_variant_sweep(50)[0].expanded_tables(oracle_ctx)
    assert all(np.array_equal(np.ascontiguousarray(getattr(a, k)).view(np.uint64), np.ascontiguousarray(getattr(b, k)).view(np.uint64))
               for k in ("members", "stations", "caps"))
