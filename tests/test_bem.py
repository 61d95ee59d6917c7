"""Potential-flow (BEM) coefficient ingestion and excitation.

Goldens (tests/golden/bem_oc3spar.npz) come from the LIVE reference: its own FOWT.readHydro ran on top of
raft_amd/bem.py's WAMIT parsers (registered as the pyhams stub -- pyHAMS is absent, so the parsers themselves are
"parity unpinned"), followed by its own BEM-excitation block with heading interpolation and solveDynamics with
frequency-dependent A_BEM / B_BEM.  tests/golden/bem/synth.{1,3} is the committed synthetic deck."""
import os

import numpy as np
import pytest

from raft_amd import bem, dropin
from raft_amd import snapshot as standin
from tests.util import group_rel_err, rel_err, case_from_fixture, load_model_fixture, random_strips, random_matrices, \
    synthetic_cases

STEM = os.path.join(standin.GOLDEN_DIR, "bem", "synth")


class _Node:
    def __init__(self, r0, r):
        self.r0, self.r = np.asarray(r0, float), np.asarray(r, float)


def test_wamit_write_read_round_trip(tmp_path):
    rng = np.random.default_rng(4)
    w = np.linspace(0.2, 2.0, 7)
    A, B = rng.normal(size=(6, 6, 7)), rng.normal(size=(6, 6, 7))
    A0, Ainf = rng.normal(size=(6, 6)), rng.normal(size=(6, 6))
    X = rng.normal(size=(3, 6, 7)) + 1j * rng.normal(size=(3, 6, 7))
    stem = str(tmp_path / "rt")
    bem.write_wamit1(stem + ".1", w, A, B, A0=A0, Ainf=Ainf)
    bem.write_wamit3(stem + ".3", w, [0.0, 90.0, 200.0], X)
    A2, B2, w1 = bem.read_wamit1(stem + ".1", TFlag=True)
    assert w1[0] == 0.0 and np.isinf(w1[1]) and np.allclose(w1[2:], w, rtol=1e-6)     # zero / infinite frequency first
    assert np.allclose(A2[:, :, 0], A0, rtol=1e-6) and np.allclose(A2[:, :, 1], Ainf, rtol=1e-6)
    assert np.allclose(A2[:, :, 2:], A, rtol=1e-6, atol=1e-6) and np.allclose(B2[:, :, 2:], B, rtol=1e-6, atol=1e-6)
    assert not np.any(B2[:, :, :2])
    M, P, R, I, w3, heads = bem.read_wamit3(stem + ".3", TFlag=True)
    assert heads == [0.0, 90.0, 200.0] and np.allclose(w3, w, rtol=1e-6)
    assert np.allclose(R + 1j * I, X, rtol=1e-6, atol=1e-6) and np.allclose(M, np.abs(X), rtol=1e-6, atol=1e-6)


def test_read_hydro_mirror_equals_reference_readHydro():
    """raft_amd.bem.read_hydro against A_BEM / B_BEM / X_BEM / BEM_headings left by the live reference's readHydro."""
    fx, model = load_model_fixture("bem_oc3spar.npz")
    f = model.fowtList[0]
    f.nodeList = [_Node(np.zeros(6), np.zeros(6))]
    f.reducedDOF = [(0, 0)]
    got = bem.read_hydro(f, STEM)
    assert np.array_equal(got.BEM_headings, fx["BEM_headings"])
    assert rel_err(got.A_BEM, fx["A_BEM"]) < 1e-13
    assert rel_err(got.B_BEM, fx["B_BEM"]) < 1e-13
    assert rel_err(got.X_BEM, fx["X_BEM"]) < 1e-13


def _bem_inputs(fx, f):
    return np.asarray(fx["BEM_headings"]), np.asarray(fx["X_BEM"])[None, :, :6, :]


def check_bem_excitation(ctx, tol):
    """raftx_bem_excitation against the live reference's F_BEM (interior, wrap-around and two-heading sea states)."""
    fx, model = load_model_fixture("bem_oc3spar.npz")
    f = model.fowtList[0]
    heads, X = _bem_inputs(fx, f)
    from raft_amd.strips import pack_fowt
    tab = pack_fowt(f)
    Z = np.zeros((1, 6, 6))
    for c in fx["cases"]:
        u = c["units"][0]
        nH = len(u["beta"])
        ctx.upload_designs([tab], Z + np.eye(6), Z, Z + np.eye(6), f.nw)
        ctx.upload_cases(f.w, f.k, f.depth, f.rho_water, f.g, np.asarray(u["zeta"])[None], np.asarray(u["beta"])[None])
        F = ctx.bem_excitation(heads, X, fetch=True)
        assert F.shape == (1, 1, nH, 6, f.nw)
        assert rel_err(F[0, 0], np.asarray(u["F_BEM"])[:, :6, :]) < tol


def test_oracle_bem_excitation_against_live_reference(oracle_ctx):
    check_bem_excitation(oracle_ctx, 1e-13)


def check_bem_sweep(ctx, tol):
    """A potential-flow deck end to end on the batched path: A_BEM / B_BEM as MBw, X_BEM interpolated on the device and
    used as the solves' F_extra -- against the live reference's solveDynamics (and the drop-in on the same fixture)."""
    fx, model = load_model_fixture("bem_oc3spar.npz")
    f = model.fowtList[0]
    cases = [case_from_fixture(fx["cases"][0]), case_from_fixture(fx["cases"][2])]
    sweep = dropin.sweep_from_models([model], cases)
    assert sweep.MBw is not None                                   # frequency-dependent added mass / damping present
    heads, X = _bem_inputs(fx, f)
    out = sweep.set_bem(heads, X).run(ctx)
    for j, ci in enumerate((0, 2)):
        c = fx["cases"][ci]
        assert int(out["niter"][0, j]) == int(c["units"][0]["niter"])
        assert group_rel_err(out["Xi"][0, j, :1], np.asarray(c["Xi"])[:1]) < tol
    # the drop-in (host F_BEM) on the two-heading case
    c = fx["cases"][1]
    eng = dropin.Engine(ctx)
    Xi = eng.solveDynamics(model, case_from_fixture(c))
    assert group_rel_err(Xi[:2], np.asarray(c["Xi"])[:2]) < tol
    assert rel_err(f.F_BEM, np.asarray(c["units"][0]["F_BEM"])) < 1e-12


def test_oracle_bem_deck_against_live_reference(oracle_ctx):
    check_bem_sweep(oracle_ctx, 1e-9)


def test_bem_excitation_argument_errors(oracle_ctx):
    from raft_amd._abi import RaftxError
    rng = np.random.default_rng(0)
    oracle_ctx.upload_designs([random_strips(rng, 5)], *random_matrices(rng, 1)[:3], 20)
    w, k, zeta, beta = synthetic_cases(rng, 1, 1, 20)
    with pytest.raises((RaftxError, ValueError)):                   # no sea states yet
        oracle_ctx.bem_excitation([0.0, 90.0], np.zeros((1, 2, 6, 20), dtype=complex))
    oracle_ctx.upload_cases(w, k, 200.0, 1025.0, 9.81, zeta, beta)
    with pytest.raises(ValueError):
        oracle_ctx.bem_excitation([0.0, 90.0], np.zeros((1, 3, 6, 20), dtype=complex))


# ------------------------------------------------------------------ GPU
@pytest.mark.gpu
def test_hip_bem_excitation_against_live_reference(hip_ctx):
    check_bem_excitation(hip_ctx, 1e-12)


@pytest.mark.gpu
def test_hip_bem_deck_against_live_reference(hip_ctx):
    check_bem_sweep(hip_ctx, 1e-9)


@pytest.mark.gpu
def test_hip_bem_excitation_batch_matches_oracle(hip_ctx, oracle_ctx):
    """Several designs with their own coefficient tables, heading adjustments and array positions, several sea states
    and headings (including exact hits on BEM headings and the 0/360 seam), plus an added force."""
    rng = np.random.default_rng(77)
    nD, nC, nH, nw, nHB = 3, 4, 2, 90, 6
    tables = [random_strips(rng, 9) for _ in range(nD)]
    mats = random_matrices(rng, nD)
    w, k, zeta, beta = synthetic_cases(rng, nC, nH, nw)
    beta[0] = np.deg2rad([45.0, 0.0])
    beta[1] = np.deg2rad([359.5, -0.25])
    heads = np.array([0.0, 45.0, 100.0, 180.0, 225.0, 300.0])
    X = rng.normal(size=(nD, nHB, 6, nw)) + 1j * rng.normal(size=(nD, nHB, 6, nw))
    hadj = np.array([0.0, 90.0, 270.0])
    xy = np.array([[0.0, 0.0], [1600.0, 0.0], [-800.0, 1200.0]])
    Fadd = rng.normal(size=(nD, nC, nH, 6, nw)) + 1j * rng.normal(size=(nD, nC, nH, 6, nw))
    out = []
    for ctx in (hip_ctx, oracle_ctx):
        ctx.upload_designs(tables, mats[0], mats[1], mats[2], nw)
        ctx.upload_cases(w, k, 200.0, 1025.0, 9.81, zeta, beta)
        a = ctx.bem_excitation(heads, X, heading_adjust=hadj, xy_ref=xy, F_add=Fadd, fetch=True)
        b = ctx.solve_dynamics(4, 0.01, 0.1)                         # uses the resident F_BEM as F_extra
        out.append((a, b))
    assert rel_err(out[0][0], out[1][0]) < 1e-12
    assert np.array_equal(out[0][1]["niter"], out[1][1]["niter"])
    assert group_rel_err(out[0][1]["Xi"].reshape(-1, 6, nw), out[1][1]["Xi"].reshape(-1, 6, nw)) < 1e-9
    # passing the same force explicitly gives the same responses
    hip_ctx.upload_cases(w, k, 200.0, 1025.0, 9.81, zeta, beta)     # drops the resident F_BEM
    c = hip_ctx.solve_dynamics(4, 0.01, 0.1, F_extra=out[0][0])
    assert group_rel_err(c["Xi"].reshape(-1, 6, nw), out[0][1]["Xi"].reshape(-1, 6, nw)) < 1e-12
