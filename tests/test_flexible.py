"""Units with more than 6 reduced DOFs (flexible members): the reference's VolturnUS-S-flexible deck (beam pontoons and
tower, 150 reduced / 360 full DOFs; tests/test_fowt.py:18-24 lists it beside the rigid decks).

  * raftx_solve_dense (the nDOF x nDOF impedance solve of every bin) against numpy.linalg.solve;
  * the node-by-node strip path (raft_amd/dropin.py Engine._excitation_general / _linearization_general) against the
    reference's OWN hydroLinearization golden of the deck and live excitation vectors;
  * Engine._solve_general against live Model.solveDynamics outputs (tests/golden/flex_volturnus.npz, written by
    oracle/make_golden.py flexible);
  * with the reference tree present: the drop-in installed into the live package, on live objects.

CPU tests run the oracle library, the ``gpu`` ones the HIP library through the same C-ABI.  Tolerances: 1e-9 on the
strip quantities; 1e-8 on the responses, whose 150 x 150 impedance matrices (FE beam stiffness beside hydrodynamic
terms) are ill-conditioned enough that two correct LU orders differ at 1e-11."""
import copy
import os

import numpy as np
import pytest

from raft_amd import dropin
from tests.util import rel_err, case_from_fixture, load_model_fixture, ref_headings


def _dense_problem(rng, n, nR, nw, mask):
    w = np.linspace(0.2, 1.4, nw)
    M = rng.normal(size=(n, n) + ((nw,) if mask & 1 else ())) + (4 * np.eye(n)[:, :, None] if mask & 1 else 4 * np.eye(n))
    B = rng.normal(size=(n, n) + ((nw,) if mask & 2 else ()))
    C = 3.0 * rng.normal(size=(n, n))
    F = rng.normal(size=(nR, n, nw)) + 1j * rng.normal(size=(nR, n, nw))
    return w, M, B, C, F


def _check_dense(ctx, tol=1e-10):
    rng = np.random.default_rng(5)
    for n, nR, nw, mask in ((1, 1, 3, 0), (7, 2, 5, 1), (37, 3, 4, 2), (97, 1, 3, 0), (128, 2, 2, 1), (150, 2, 6, 3), (158, 2, 2, 0),
                            (159, 2, 2, 0), (257, 1, 2, 0)):
        w, M, B, C, F = _dense_problem(rng, n, nR, nw, mask)
        Xi, Z = ctx.solve_dense(w, M, B, C, F, want_Z=True)
        assert Xi.shape == (nR, n, nw) and Z.shape == (n, n, nw)
        for i in range(nw):
            Zr = -w[i] ** 2 * (M[:, :, i] if mask & 1 else M) + 1j * w[i] * (B[:, :, i] if mask & 2 else B) + C
            assert rel_err(Z[:, :, i], Zr) < 1e-15
            for r in range(nR):
                assert rel_err(Xi[r, :, i], np.linalg.solve(Zr, F[r, :, i])) < tol * max(1.0, np.linalg.cond(Zr) * 1e-3)
        assert rel_err(ctx.solve_dense(w, M, B, C, F), Xi) == 0.0            # without Z: the same responses
    with pytest.raises(ValueError):
        ctx.solve_dense(np.ones(2), np.eye(3), np.eye(3), np.eye(3), np.ones((1, 4, 2)))


def _check_strips(ctx):
    fx, model = load_model_fixture("refgold_VolturnUS-S-flexible.npz")
    eng = dropin.Engine(ctx)
    fowt = model.fowtList[0]
    assert fowt.nDOF == 150 and fowt.nFullDOF == 360
    for i, c in enumerate(fx["exc_cases"]):
        eng.calcHydroExcitation(fowt, dict(c), memberList=fowt.memberList)
        assert rel_err(fowt.F_hydro_iner, fx["exc_F_hydro_iner"][i]) < 1e-9
        assert rel_err(fowt.F_hydro_iner_fullDOF, fx["exc_F_hydro_iner_fullDOF"][i]) < 1e-9
        assert fowt.F_BEM.shape == fowt.F_hydro_iner.shape and not np.any(fowt.F_BEM)
    # the reference's own test, tests/test_fowt.py:150-175
    case = {'wave_spectrum': 'unit', 'wave_heading': 0, 'wave_period': 10, 'wave_height': 2}
    eng.calcHydroExcitation(fowt, case, memberList=fowt.memberList)
    phase = np.linspace(0, 2 * np.pi, fowt.nw * fowt.nDOF).reshape(fowt.nDOF, fowt.nw)
    Xi = 0.1 * np.exp(1j * phase)
    B = eng.calcHydroLinearization(fowt, Xi)
    F = eng.calcDragExcitation(fowt, 0)
    assert B.shape == (150, 150) and F.shape == (150, fowt.nw)
    np.testing.assert_allclose(B, fx["lin_B_hydro_drag"], rtol=1e-5, atol=1e-10)     # the reference's gate
    np.testing.assert_allclose(F, fx["lin_F_hydro_drag"], rtol=1e-5)
    assert rel_err(B, fx["lin_B_hydro_drag"]) < 1e-9
    assert rel_err(F, fx["lin_F_hydro_drag"]) < 1e-9
    # a member list that leaves members out: only their nodes' rows carry excitation (raft_fowt.py:1854-1857)
    eng.calcHydroExcitation(fowt, case, memberList=fowt.memberList[:1])
    node0 = fowt.memberList[0].nodeList[0].id * 6
    rows = np.abs(fowt.F_hydro_iner_fullDOF).max(axis=(0, 2)) > 0
    assert rows[node0:node0 + 6].any() and not rows[:node0].any() and not rows[node0 + 6:].any()


def _check_solve(ctx):
    fx, model = load_model_fixture("flex_volturnus.npz")
    eng = dropin.Engine(ctx)
    fowt = model.fowtList[0]
    for c in fx["cases"]:
        Xi = eng.solveDynamics(model, case_from_fixture(c))
        Xr, nH = ref_headings(c)
        assert Xi.shape == (nH + 1, 150, model.nw) and np.all(Xi[nH] == 0)
        assert rel_err(Xi[:nH], Xr) < 1e-8
        assert rel_err(Xi[:nH, :6], Xr[:, :6]) < 1e-8                        # the rigid-body rows on their own scale
        u = c["units"][0]
        assert int(model._raftx_niter[0]) == int(u["niter"])      # the deck's nIter = 4 ends unconverged, as upstream
        assert rel_err(fowt.B_hydro_drag, u["B_hydro_drag"]) < 1e-9
        assert rel_err(fowt.Xi_fullDOF[:nH], c["Xi_fullDOF"]) < 1e-8
        assert fowt.Z.shape == (150, 150, model.nw)
    c = fx["case_converged"]                                                 # more iterations allowed: converges, as upstream
    model.nIter = int(fx["nIter_converged"])
    Xi = eng.solveDynamics(model, case_from_fixture(c))
    assert rel_err(Xi[:1], ref_headings(c)[0]) < 1e-8
    assert int(model._raftx_niter[0]) == int(c["units"][0]["niter"]) < 16 and int(model._raftx_flags[0]) == 1
    with pytest.raises(dropin.UnsupportedFOWT):                              # outputs need the live object's nodes / members (the
        eng.saveTurbineOutputs(fowt, {}, case_from_fixture(fx["cases"][0]))  # stand-in has none): tests/test_dropin_live_reference.py


def _check_flex_sweep(ctx, n_unit):
    """raft_amd.flex.FlexSweep: n_unit copies of the flexible deck x three sea states in one batch (node sweeps of the whole
    batch in one launch per iteration, every impedance solve of an iteration in one launch) equal the drop-in's one-case-
    at-a-time solveDynamics -- responses, iteration counts, convergence flags, B_hydro_drag -- and the live reference."""
    fx, model = load_model_fixture("flex_volturnus.npz")
    eng = dropin.Engine(ctx)
    base = case_from_fixture(fx["cases"][0])
    cases = [base, dict(base, wave_height=4.0, wave_period=9.0, wave_heading=-20.0), dict(base, wave_height=1.0, wave_period=6.0)]
    single, nit, flg, Bd = [], [], [], []
    for c in cases:
        single.append(eng.solveDynamics(model, dict(c)).copy())
        nit.append(int(model._raftx_niter[0]))
        flg.append(int(model._raftx_flags[0]))
        Bd.append(np.array(model.fowtList[0].B_hydro_drag))
    sw = dropin.flex_sweep_from_models([model] * n_unit, cases)
    out = sw.run(ctx, want_Z=(n_unit == 1))
    assert out["Xi"].shape == (n_unit, 3, 1, 150, model.nw)
    for d in range(n_unit):
        assert list(out["niter"][d]) == nit and list(out["flags"][d] & 1) == flg
        for ic in range(3):
            assert rel_err(out["Xi"][d, ic, 0], single[ic][0]) < 1e-10
            assert rel_err(out["B_drag"][d, ic], Bd[ic]) < 1e-12
    assert rel_err(out["Xi"][0, 0, :1], ref_headings(fx["cases"][0])[0][:1]) < 1e-8
    if n_unit == 1:
        eng.solveDynamics(model, dict(cases[2]))
        assert rel_err(out["Z"][0, 2], model.fowtList[0].Z) < 1e-12
    with pytest.raises(dropin.UnsupportedFOWT):             # rigid units go through the 6-DOF sweeps
        dropin.flex_sweep_from_models([load_model_fixture("c1_oc3spar.npz")[1]], cases)


def _flex_problem(rng, n_unit, nodes_per_unit, n, n_case, n_head, freq_dep):
    """A synthetic batch for raftx_flex_solve: node strip tables taken from the flexible deck (the first nodes of its
    packing, re-used unit after unit), random T rows, well-conditioned M / B / C, random excitation."""
    fx, model = load_model_fixture("flex_volturnus.npz")
    fowt = model.fowtList[0]
    from raft_amd.strips import pack_fowt_nodes
    _, tables = pack_fowt_nodes(fowt, fowt.memberList)
    node_off = np.concatenate([[0], np.cumsum(nodes_per_unit)]).astype(np.int64)
    tabs = [tables[(3 * u + i) % len(tables)] for u in range(n_unit) for i in range(nodes_per_unit[u])]
    Tn = rng.normal(size=(int(node_off[-1]), 6, n)) * 0.3
    nw = int(model.nw)                                     # (the node tables carry the deck's MacCamy-Fuchs columns: its grid)
    w = np.asarray(model.w)
    k = np.asarray(fowt.k)
    sh = (n_unit, n, n, nw) if freq_dep else (n_unit, n, n)
    M = rng.normal(size=sh) * 1e5 + (np.eye(n) * 2e7)[(None, ..., None) if freq_dep else (None, ...)]
    B = rng.normal(size=(n_unit, n, n)) * 1e4 + np.eye(n)[None] * 1e6
    C = rng.normal(size=(n_unit, n, n)) * 1e5 + np.eye(n)[None] * 3e7
    zeta = np.abs(rng.normal(size=(n_case, n_head, nw))) * np.array([0.05, 1.5, 0.4][:n_case])[:, None, None]
    beta = rng.uniform(-0.5, 0.5, size=(n_case, n_head))
    F_lin = (rng.normal(size=(n_unit, n_case, n_head, n, nw)) + 1j * rng.normal(size=(n_unit, n_case, n_head, n, nw))) * 1e6
    return dict(node_off=node_off, tabs=tabs, Tn=Tn, w=w, k=k, depth=float(fowt.depth), M=M, B=B, C=C, zeta=zeta, beta=beta, F_lin=F_lin)


def _run_flex(ctx, P, units=None, cases=None, nIter=12, tol=0.01):
    """raftx_flex_solve on (a subset of) the problem's units and sea states"""
    us = list(range(len(P["node_off"]) - 1)) if units is None else units
    cs = list(range(P["zeta"].shape[0])) if cases is None else cases
    nodes = [i for u in us for i in range(int(P["node_off"][u]), int(P["node_off"][u + 1]))]
    off = np.concatenate([[0], np.cumsum([int(P["node_off"][u + 1] - P["node_off"][u]) for u in us])])
    nw = len(P["w"])
    Z6 = np.zeros((len(nodes), 6, 6))
    ctx.upload_designs([P["tabs"][i] for i in nodes], Z6, Z6, Z6, nw)
    ctx.upload_cases(P["w"], P["k"], P["depth"], 1025.0, 9.81, P["zeta"][cs], P["beta"][cs])
    return ctx.flex_solve(off, P["Tn"][nodes], P["M"][us], P["B"][us], P["C"][us], P["F_lin"][us][:, cs], nIter, tol, 0.1, want_Z=True)


def _check_flex_solve(ctx, other=None):
    """raftx_flex_solve directly: ragged units (3, 1 and 5 wet nodes), two headings, three sea states that need different
    numbers of iterations, frequency-dependent M.  A pair's result is that of solving it ALONE (converged pairs are frozen,
    the others go on); `other`: a second backend that must agree (device against the checker)."""
    rng = np.random.default_rng(11)
    P = _flex_problem(rng, 3, [3, 1, 5], 20, 3, 2, True)
    out = _run_flex(ctx, P)
    assert out["Xi"].shape == (3, 3, 2, 20, len(P["w"])) and np.isfinite(out["Xi"]).all()
    assert len(set(out["niter"].ravel().tolist())) > 1, out["niter"]          # the batch really has pairs of different length
    assert (out["flags"] == 1).all()
    for u, c in ((0, 0), (1, 2), (2, 1)):
        alone = _run_flex(ctx, P, [u], [c])
        assert int(alone["niter"][0, 0]) == int(out["niter"][u, c])
        for key in ("Xi", "B_drag", "F_drag", "Z"):
            assert rel_err(alone[key][0, 0], out[key][u, c]) < 1e-12, key
    Zchk = -(P["w"] ** 2) * P["M"][2] + 1j * P["w"] * (P["B"][2] + out["B_drag"][2, 1])[:, :, None] + P["C"][2][:, :, None]
    assert rel_err(out["Z"][2, 1], Zchk) < 1e-13
    for h in range(2):                                                        # Z Xi = F_lin + F_drag, every heading
        lhs = np.einsum("ijw,jw->iw", out["Z"][2, 1], out["Xi"][2, 1, h])
        assert rel_err(lhs, P["F_lin"][2, 1, h] + out["F_drag"][2, 1, h]) < 1e-9
    few = _run_flex(ctx, P, nIter=1)                                          # the iteration cap: not converged, flagged so
    assert (few["niter"] == 2).all() and not (few["flags"] & 1).any()
    if other is not None:
        ref = _run_flex(other, P)
        assert np.array_equal(ref["niter"], out["niter"]) and np.array_equal(ref["flags"], out["flags"])
        for key in ("Xi", "B_drag", "F_drag", "Z"):
            assert rel_err(out[key], ref[key]) < 1e-9, key
    with pytest.raises(Exception, match="nodeOff"):
        ctx.flex_solve([0, 2], P["Tn"][:2], P["M"][:1], P["B"][:1], P["C"][:1], P["F_lin"][:1], 3, 0.01, 0.1)


def test_oracle_flex_solve_entry(oracle_ctx):
    _check_flex_solve(oracle_ctx)


@pytest.mark.gpu
def test_hip_flex_solve_entry(hip_ctx, oracle_ctx):
    _check_flex_solve(hip_ctx, oracle_ctx)


@pytest.mark.gpu
def test_hip_flex_solve_larger_systems(hip_ctx, oracle_ctx):
    """150 reduced DOFs (the register-resident dense kernel's largest grid) and 170 (the L2-workspace kernel), one heading"""
    for n in (150, 170):
        P = _flex_problem(np.random.default_rng(n), 2, [4, 2], n, 2, 1, False)
        a, b = _run_flex(hip_ctx, P), _run_flex(oracle_ctx, P)
        assert np.array_equal(a["niter"], b["niter"]) and np.array_equal(a["flags"], b["flags"])
        assert rel_err(a["Xi"], b["Xi"]) < 1e-9 and rel_err(a["B_drag"], b["B_drag"]) < 1e-11


@pytest.mark.gpu
def test_hip_flex_solve_iterations_as_a_graph(hip_ctx, tmp_path):
    """RAFTX_FLEX_GRAPH=1: the launches of an iteration captured once into a hipGraph and replayed (in a process of its own: the
    setting is read once) -- the same bits as plain launches."""
    import os
    import subprocess
    import sys
    out = str(tmp_path / "graph.npz")
    code = ("import numpy as np, tests.test_flexible as t; from raft_amd import backend; ctx = backend.hip_library().context(0); "
            "P = t._flex_problem(np.random.default_rng(11), 3, [3, 1, 5], 20, 3, 2, True); r = t._run_flex(ctx, P); "
            "np.savez(%r, Xi=r['Xi'], niter=r['niter'], B=r['B_drag'])" % out)
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, RAFTX_FLEX_GRAPH="1"), capture_output=True, text=True, timeout=600,
                       cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    got = np.load(out)
    ref = _run_flex(hip_ctx, _flex_problem(np.random.default_rng(11), 3, [3, 1, 5], 20, 3, 2, True))
    assert int(got["niter"].max()) > 2                                       # (replays happened)
    assert np.array_equal(got["niter"], ref["niter"])
    assert np.array_equal(got["Xi"].view(np.float64), ref["Xi"].view(np.float64)) and np.array_equal(got["B"], ref["B_drag"])


def _check_flex_gemm(ctx):
    rng = np.random.default_rng(4)
    for K, n in ((360, 150), (6, 7), (54, 33), (12, 16)):
        A, W = rng.normal(size=(K, n)), rng.normal(size=(K, n))
        got = ctx.debug_flex_gemm(A, W)
        assert rel_err(got, A.T @ W) < 1e-13 and rel_err(got, (A.T @ W).T) > 0.1       # (an asymmetric product)


def test_oracle_flex_gemm(oracle_ctx):
    _check_flex_gemm(oracle_ctx)


@pytest.mark.gpu
def test_hip_flex_projection_gemm_tiles(hip_ctx):
    """the MFMA tiles of sum_nodes T^T B T against NumPy, with an asymmetric product"""
    _check_flex_gemm(hip_ctx)


def test_flexible_sweep_keeps_its_host_arrays(oracle_lib):
    """FlexSweep holds its host arrays (page-locked on the device backend) across runs, in the context's store: a second run
    re-uses them and gives the same bits; copy=False hands out views of them; release() / a new context start afresh."""
    fx, model = load_model_fixture("flex_volturnus.npz")
    base = case_from_fixture(fx["cases"][0])
    sw = dropin.flex_sweep_from_models([model], [base])
    ctx = oracle_lib.context(0)
    a = sw.run(ctx)
    v1 = sw.run(ctx, copy=False)
    v2 = sw.run(ctx, copy=False)
    assert v1["Xi"] is not a["Xi"] and np.shares_memory(v1["Xi"], v2["Xi"]) and not np.shares_memory(a["Xi"], v1["Xi"])
    assert np.array_equal(a["Xi"].view(np.float64), v2["Xi"].view(np.float64)) and np.array_equal(a["B_drag"], v2["B_drag"])
    assert len(ctx._flex_bufs) == 1
    sw.release(ctx)
    assert len(ctx._flex_bufs) == 0
    b = sw.run(ctx)                                                          # allocates again
    assert np.array_equal(a["Xi"].view(np.float64), b["Xi"].view(np.float64))
    ctx.close()
    ctx2 = oracle_lib.context(0)                                             # another context: its own arrays
    c = sw.run(ctx2)
    assert np.array_equal(a["Xi"].view(np.float64), c["Xi"].view(np.float64))
    ctx2.close()


def test_oracle_flexible_sweep(oracle_ctx):
    _check_flex_sweep(oracle_ctx, 1)


@pytest.mark.gpu
def test_hip_flexible_sweep(hip_ctx):
    _check_flex_sweep(hip_ctx, 3)


@pytest.mark.gpu
def test_hip_solve_dense_batch_equals_single_calls(hip_ctx, oracle_ctx):
    """raftx_solve_dense_batch: five 150-DOF systems (register-resident kernel) and five 40-DOF ones (L2-workspace kernel)
    in one launch each equal the single-system calls bit for bit, and the oracle to rounding."""
    rng = np.random.default_rng(5)
    for n in (150, 40):
        probs = [_dense_problem(rng, n, 2, 6, 0) for _ in range(5)]
        w = probs[0][0]
        M, B, C, F = (np.array([p[i] for p in probs]) for i in (1, 2, 3, 4))
        Xb, Zb = hip_ctx.solve_dense_batch(w, M, B, C, F, want_Z=True)
        for i in range(5):
            Xs, Zs = hip_ctx.solve_dense(w, M[i], B[i], C[i], F[i], want_Z=True)
            assert np.array_equal(Xb[i].view(np.uint64), Xs.view(np.uint64)) and np.array_equal(Zb[i].view(np.uint64), Zs.view(np.uint64))
        Xo = oracle_ctx.solve_dense_batch(w, M, B, C, F)
        assert rel_err(Xb, Xo) < 1e-10


def _check_dense_resident(ctx):
    """raftx_dense_resident / raftx_solve_dense_resident: the matrices of two units kept on the device, three systems per unit
    (its sea states) that add their own Badd to B -- equal to the stateless batch call with B + Badd written out; for the
    register-resident kernel's shapes and the L2-workspace one's, frequency-dependent M / B and not."""
    rng = np.random.default_rng(8)
    for n, nR, nw, mask in ((12, 1, 4, 0), (40, 2, 3, 2), (150, 1, 5, 3), (170, 2, 3, 1)):
        probs = [_dense_problem(rng, n, nR, nw, mask) for _ in range(2)]
        w = probs[0][0]
        M, B, C = (np.array([p[i] for p in probs]) for i in (1, 2, 3))
        F = rng.normal(size=(6, nR, n, nw)) + 1j * rng.normal(size=(6, nR, n, nw))
        Badd = rng.normal(size=(6, n, n))
        ctx.dense_resident(w, M, B, C)
        X, Z = ctx.solve_dense_resident(F, Badd=Badd, want_Z=True)
        X0 = ctx.solve_dense_resident(F)                                      # no Badd: B alone
        Bfull = np.repeat(B, 3, axis=0) + (Badd[..., None] if mask & 2 else Badd)
        Xb, Zb = ctx.solve_dense_batch(w, np.repeat(M, 3, axis=0), Bfull, np.repeat(C, 3, axis=0), F, want_Z=True)
        assert np.array_equal(X.view(np.float64), Xb.view(np.float64)) and np.array_equal(Z.view(np.float64), Zb.view(np.float64))
        assert np.array_equal(X0.view(np.float64), ctx.solve_dense_batch(w, np.repeat(M, 3, axis=0), np.repeat(B, 3, axis=0),
                                                                         np.repeat(C, 3, axis=0), F).view(np.float64))
        ctx.dense_resident(None, None, None, None)
        with pytest.raises(Exception, match="resident"):
            ctx.solve_dense_resident(F)


def test_oracle_solve_dense_resident(oracle_ctx):
    _check_dense_resident(oracle_ctx)


@pytest.mark.gpu
def test_hip_solve_dense_resident(hip_ctx):
    _check_dense_resident(hip_ctx)


def test_oracle_solve_dense(oracle_ctx):
    _check_dense(oracle_ctx)


def _check_flexible_mcf(ctx):
    """MacCamy-Fuchs on FLEXIBLE members (VERDICT r5 missing 4; raft_member.py:1415-1420 with the node-by-node sums of
    :1969-1976): the flexible deck with its three MacCamy-Fuchs outer columns as beam members (240 reduced DOFs; fixture by
    oracle/make_golden.py flexmcf from the live reference).  Every wet node of such a column is a one-strip table that carries
    its own row of the complex Cm table: calcHydroExcitation and the whole solveDynamics against the reference's."""
    fx, model = load_model_fixture("flex_mcf.npz")
    fowt = model.fowtList[0]
    assert any(getattr(m, "type", "rigid") != "rigid" and getattr(m, "MCF", False) for m in fowt.memberList)
    eng = dropin.Engine(ctx)
    for c, F_ref in zip(fx["exc_cases"], fx["exc_F_hydro_iner"]):
        eng.calcHydroExcitation(fowt, dict(c), memberList=fowt.memberList)
        assert fowt.F_hydro_iner.shape == F_ref.shape and rel_err(fowt.F_hydro_iner, F_ref) < 1e-12
    c = fx["case"]
    Xi = eng.solveDynamics(model, case_from_fixture(c))
    Xr, nH = ref_headings(c)
    assert Xi.shape[1] == fowt.nDOF == 240 and rel_err(Xi[:nH], Xr) < 1e-8 and rel_err(Xi[:nH, :6], Xr[:, :6]) < 1e-8
    assert int(model._raftx_niter[0]) == int(c["units"][0]["niter"])
    assert rel_err(fowt.B_hydro_drag, c["units"][0]["B_hydro_drag"]) < 1e-9


def test_oracle_flexible_members_with_maccamy_fuchs(oracle_ctx):
    _check_flexible_mcf(oracle_ctx)


@pytest.mark.gpu
def test_hip_flexible_members_with_maccamy_fuchs(hip_ctx):
    _check_flexible_mcf(hip_ctx)


def test_oracle_flexible_strips(oracle_ctx):
    _check_strips(oracle_ctx)


def test_oracle_flexible_solveDynamics(oracle_ctx):
    _check_solve(oracle_ctx)


def test_unsupported_flexible_variants(oracle_ctx):
    fx, model = load_model_fixture("flex_volturnus.npz")
    eng = dropin.Engine(oracle_ctx)
    case = case_from_fixture(fx["cases"][0])
    fowt = model.fowtList[0]
    fowt.potSecOrder = 1
    with pytest.raises(dropin.UnsupportedFOWT):
        eng.solveDynamics(model, copy.deepcopy(case))
    fowt.potSecOrder = 0
    model.fowtList = [fowt, fowt]
    with pytest.raises(dropin.UnsupportedFOWT):
        eng.solveDynamics(model, copy.deepcopy(case))


def _with_bem(model, seed=4):
    """potential-flow coefficients lumped at the first six DOFs of the 150-DOF unit (raft_fowt.py:1479-1501)"""
    rng = np.random.default_rng(seed)
    f = model.fowtList[0]
    n, nw, w = int(f.nDOF), model.nw, np.asarray(model.w)
    sym = lambda a: 0.5 * (a + a.T)
    f.potModMaster = 2
    f.heading_adjust = getattr(f, "heading_adjust", 0.0)
    f.BEM_headings = np.array([0.0, 120.0, 240.0])
    f.A_BEM, f.B_BEM = np.zeros([n, n, nw]), np.zeros([n, n, nw])
    sc = np.outer([1e3, 1e3, 1e3, 1e4, 1e4, 1e4], [1e3, 1e3, 1e3, 1e4, 1e4, 1e4])
    f.A_BEM[:6, :6] = ((sym(rng.uniform(0, 1, (6, 6))) + 2 * np.eye(6)) * sc)[:, :, None] / (1.0 + (w / 0.8) ** 2)
    f.B_BEM[:6, :6] = ((sym(rng.uniform(0, 1, (6, 6))) + np.eye(6)) * sc)[:, :, None] * ((w / 0.6) / (1.0 + (w / 0.6) ** 2))
    f.X_BEM = np.zeros([3, n, nw], dtype=complex)
    f.X_BEM[:, :6] = np.array([2e6, 2e6, 1e6, 4e7, 4e7, 1e7])[None, :, None] * np.exp(1j * (rng.uniform(0, 6, (3, 6, 1)) + 2.0 * w))
    return f


def _check_flexible_with_bem(ctx, other=None):
    """A unit with 150 reduced DOFs AND potential-flow coefficients (the NumPy path is compared on live objects in
    tests/test_dropin_live_reference.py): the coefficients change the response, F_BEM is T^T of the first six full-DOF rows,
    and a second backend gives the same numbers."""
    fx, model = load_model_fixture("flex_volturnus.npz")
    case = case_from_fixture(fx["cases"][0])
    Xi0 = dropin.Engine(ctx).solveDynamics(model, copy.deepcopy(case)).copy()
    f = _with_bem(model)
    Xi1 = dropin.Engine(ctx).solveDynamics(model, copy.deepcopy(case)).copy()
    assert np.any(f.F_BEM) and rel_err(f.F_BEM, np.einsum("fd,hfw->hdw", np.asarray(f.T), f.F_BEM_fullDOF)) < 1e-14
    assert not np.any(f.F_BEM_fullDOF[:, 6:]) and rel_err(Xi1, Xi0) > 1e-2
    assert rel_err(f.Z[:6, :6] - (-np.asarray(model.w) ** 2 * f.A_BEM[:6, :6] + 1j * np.asarray(model.w) * f.B_BEM[:6, :6]),
                   f.Z[:6, :6]) < 2.0                                          # Z carries the coefficients (finite, same shape)
    if other is not None:
        _, m2 = load_model_fixture("flex_volturnus.npz")
        f2 = _with_bem(m2)
        Xi2 = dropin.Engine(other).solveDynamics(m2, copy.deepcopy(case)).copy()
        assert rel_err(Xi1, Xi2) < 1e-8 and rel_err(f.F_BEM, f2.F_BEM) < 1e-12 and np.array_equal(model._raftx_niter, m2._raftx_niter)


def test_oracle_flexible_unit_with_potential_flow_coefficients(oracle_ctx):
    _check_flexible_with_bem(oracle_ctx)


@pytest.mark.gpu
def test_hip_flexible_unit_with_potential_flow_coefficients(hip_ctx, oracle_ctx):
    _check_flexible_with_bem(hip_ctx, oracle_ctx)


@pytest.mark.gpu
def test_hip_solve_dense(hip_ctx):
    _check_dense(hip_ctx)


@pytest.mark.gpu
def test_hip_solve_dense_equals_oracle_and_flags_singular(hip_ctx, oracle_ctx):
    rng = np.random.default_rng(11)
    w, M, B, C, F = _dense_problem(rng, 150, 2, 8, 2)
    Xh, Zh = hip_ctx.solve_dense(w, M, B, C, F, want_Z=True)
    Xo, Zo = oracle_ctx.solve_dense(w, M, B, C, F, want_Z=True)
    assert rel_err(Zh, Zo) < 1e-15 and rel_err(Xh, Xo) < 1e-11
    Z0 = np.zeros((3, 3))
    X = hip_ctx.solve_dense(np.ones(1), Z0, Z0, Z0, np.ones((1, 3, 1)))          # singular: not finite, never garbage
    assert not np.isfinite(X).any()


@pytest.mark.gpu
def test_hip_flexible_strips(hip_ctx):
    _check_strips(hip_ctx)


@pytest.mark.gpu
def test_hip_flexible_solveDynamics(hip_ctx):
    _check_solve(hip_ctx)


def test_installed_flexible_solveDynamics_equals_numpy_path(oracle_ctx):
    """The patched package on the live flexible deck vs the reference's NumPy path (container only)."""
    from oracle import ref_harness as rh
    if not rh.tree_available():
        pytest.skip("reference tree not present")
    rh.import_raft()
    from raft import raft_model
    d = rh.prepare_design(rh.load_design(os.path.join(rh.REFERENCE_ROOT, "tests/test_data/VolturnUS-S-flexible.yaml")))
    cm = np.zeros((150, 150))
    cm[:6, :6] = rh.DEFAULT_C_MOOR
    m_new, m_old = rh.build_model(d, c_moor=cm), rh.build_model(d, c_moor=cm)
    case = rh.make_case(Hs=4.0, Tp=10.0, heading=-40.0)
    Xi_old = m_old.solveDynamics(copy.deepcopy(case)).copy()
    dropin._default_engine = dropin.Engine(oracle_ctx)
    saved = dropin.install()
    try:
        assert raft_model.Model.solveDynamics is dropin.solveDynamics
        Xi_new = m_new.solveDynamics(copy.deepcopy(case)).copy()
        fn, fo = m_new.fowtList[0], m_old.fowtList[0]
        assert rel_err(Xi_new, Xi_old) < 1e-8
        assert rel_err(fn.Z, fo.Z) < 1e-12
        assert rel_err(fn.B_hydro_drag, fo.B_hydro_drag) < 1e-9
        assert rel_err(fn.F_hydro_iner_fullDOF, fo.F_hydro_iner_fullDOF) < 1e-12
        assert rel_err(fn.Xi_fullDOF, fo.Xi_fullDOF) < 1e-8
        # the FOWT-level methods on the live object (raft_fowt.py:1732, 1891, 1940)
        fo.calcHydroExcitation(copy.deepcopy(case), memberList=fo.memberList)    # patched method, live object
        Bn = fo.calcHydroLinearization(fn.Xi[0])
        Fn = fo.calcDragExcitation(0)
    finally:
        dropin.uninstall(saved)
        dropin._default_engine = dropin.Engine()
    Bo = fo.calcHydroLinearization(fn.Xi[0])                                    # the reference's own methods again
    Fo = fo.calcDragExcitation(0)
    assert rel_err(Bn, Bo) < 1e-10 and rel_err(Fn, Fo) < 1e-10


def _check_flexible_dynamic_mooring(ctx):
    """A unit with more than 6 reduced DOFs and its own lumped-mass mooring (moorMod == 2, raft_model.py:1019-1030,1069-1072):
    the mooring model's M, A, C lumped at the first six reduced DOFs about XiStart, its damping re-linearised about EVERY
    iterate -- the drop-in steps the device fixed point one launch per iteration from an explicit linearisation point
    (raftx_flex_start).  Against the LIVE reference's solveDynamics on the flexible deck with the same stand-in mooring
    (tests/golden/flex_moormod2.npz, oracle/make_golden.py flexmoor): responses of every heading, the impedance of the
    last iteration, iteration counts and the number of mooring updates; the iteration cap."""
    from tests.util import attach_fake_lines_at
    from raft_amd.snapshot import load_fixture
    gold = load_fixture("flex_moormod2.npz")
    _, model = load_model_fixture("flex_volturnus.npz")
    ms = attach_fake_lines_at(model, gold["moor_arm"])
    eng = dropin.Engine(ctx)
    seen = set()
    for c in gold["cases"]:
        model.nIter = int(c["nIter"])
        ms.calls = 0
        Xi = eng.solveDynamics(model, case_from_fixture(c))
        nH = c["Xi"].shape[0]
        assert Xi.shape == (nH + 1, 150, model.nw) and not np.any(Xi[nH])
        assert int(model._raftx_niter[0]) == int(c["units"][0]["niter"])
        assert ms.calls == int(c["mooring_updates"])                          # one about XiStart + one per iteration
        assert rel_err(Xi[:nH], c["Xi"]) < 1e-8
        assert rel_err(model.fowtList[0].Z, c["Z"]) < 1e-10
        assert rel_err(model.fowtList[0].B_hydro_drag, c["units"][0]["B_hydro_drag"]) < 1e-9
        seen.add(bool(model._raftx_flags[0] & 1))
    assert seen == {True}
    model.nIter = 1                                                           # the iteration cap: two passes, flagged unconverged
    ms.calls = 0
    eng.solveDynamics(model, case_from_fixture(gold["cases"][1]))
    assert int(model._raftx_niter[0]) == 2 and not (model._raftx_flags[0] & 1) and ms.calls == 3
    # the explicit start is one-shot and shape-checked
    fowt = model.fowtList[0]
    rows, tables, Tn = fowt._raftx_nodes
    ctx.flex_start(np.zeros((2, 1, 150, model.nw), dtype=complex))
    with pytest.raises(Exception, match="linearisation point"):
        ctx.flex_solve([0, len(rows)], Tn, np.eye(150)[None], np.eye(150)[None], np.eye(150)[None],
                       np.zeros((1, 1, fowt.nWaves, 150, model.nw), dtype=complex), 0, 0.01, 0.1)


def test_oracle_flexible_unit_with_dynamic_mooring(oracle_ctx):
    _check_flexible_dynamic_mooring(oracle_ctx)


@pytest.mark.gpu
def test_hip_flexible_unit_with_dynamic_mooring(hip_ctx):
    _check_flexible_dynamic_mooring(hip_ctx)
