"""Row f4 pinned on the coefficient files the reference itself ships.

CPU suite.  The text files live in the reference tree (raft/data/cylinder/Output/Wamit_format/Buoy.{1,3},
tests/test_data/OC4semi-WAMIT_Coefs/marin_semi.{1,12d}); what the reference makes of them is committed under
tests/golden/ (oracle/make_golden.py f4): the reference's OWN golden for calcBEM -> readHydro (written upstream with
pyHAMS' parser, tests/test_fowt.py:218-241) and FOWT.readQTF run live.  Tests that need a text file are skipped where
the reference tree is absent (the GPU box); the potSecOrder == 2 solve runs from the committed arrays everywhere."""
import os

import numpy as np
import pytest

from raft_amd import bem, snapshot
from raft_amd import qtf as rq
from tests.util import rel_err, group_rel_err, load_model_fixture, case_from_fixture

REF = os.environ.get("RAFT_REFERENCE_ROOT", "/root/reference")
needs_files = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "tests", "test_data", "OC4semi-WAMIT_Coefs")),
                                 reason="reference tree not present")


@needs_files
def test_marin_semi_dot1_against_the_references_own_golden():
    """read_wamit1 + the readHydro arithmetic on marin_semi.1 reproduce A_BEM / B_BEM of
    OC4semi-WAMIT_Coefs_true_BEM_forces.pkl -- values upstream computed with pyHAMS' read_wamit1 -- to rounding
    (upstream's own gate: rtol 1e-5, atol 1e-3)."""
    g = snapshot.load_fixture("refgold_bem_oc4semi.npz")
    A, B = bem.added_mass_damping(os.path.join(REF, str(g["file"])), g["w"], float(g["rho_water"]), g["r0"])
    np.testing.assert_allclose(A, g["A_BEM"], rtol=1e-5, atol=1e-3)
    np.testing.assert_allclose(B, g["B_BEM"], rtol=1e-5, atol=1e-3)
    assert rel_err(A, g["A_BEM"]) < 1e-12 and rel_err(B, g["B_BEM"]) < 1e-12
    # the file itself: period-based, longest period first (it carries no PER = -1 / PER = 0 limiting sets: readHydro
    # uses its first set as the zero-frequency added mass and skips the second, raft_fowt.py:1469-1470)
    Af, Bf, w1 = bem.read_wamit1(os.path.join(REF, str(g["file"])), TFlag=True)
    assert Af.shape == Bf.shape == (6, 6, len(w1)) and len(w1) > 40
    assert np.all(np.diff(w1) > 0) and abs(w1[0] - 0.01) < 1e-6
    assert np.abs(Af - Af.transpose(1, 0, 2)).max() < 2e-2 * np.abs(Af).max()     # a symmetric added-mass matrix, as printed


@needs_files
def test_hams_cylinder_files_parse_consistently():
    """raft/data/cylinder/Output/Wamit_format/Buoy.1 / Buoy.3 (the HAMS example output upstream ships): every column
    lands where the format says -- the modulus / phase columns of the .3 file agree with its real / imaginary columns,
    the .1 matrices are those of a body of revolution (symmetric, surge = sway, roll = pitch), frequencies agree."""
    stem = os.path.join(REF, "raft", "data", "cylinder", "Output", "Wamit_format", "Buoy")
    A, B, f1 = bem.read_wamit1(stem + ".1")
    M, P, R, I, f3, heads = bem.read_wamit3(stem + ".3")
    assert A.shape[:2] == (6, 6) and A.shape == B.shape and A.shape[2] == len(f1) > 10
    assert np.array_equal(np.unique(f1), np.unique(f3)) and len(heads) >= 1
    scale = np.abs(A).max()
    assert np.abs(A - A.transpose(1, 0, 2)).max() < 1e-3 * scale               # a panel solution: symmetric to its own accuracy
    assert rel_err(A[0, 0], A[1, 1]) < 1e-3 and rel_err(A[3, 3], A[4, 4]) < 1e-3
    assert np.all(B[0, 0] >= -1e-12) and np.all(B[2, 2] >= -1e-12)              # radiation damping is not negative
    big = M > 1e-6 * M.max()
    assert np.allclose(M[big], np.hypot(R, I)[big], rtol=2e-6)
    ph = np.degrees(np.arctan2(I, R))
    assert np.abs(((P - ph + 180.0) % 360.0 - 180.0)[big]).max() < 2e-3        # printed with 7 significant digits


@needs_files
def test_read_qtf12d_equals_the_references_readQTF():
    """raft_amd.qtf.read_qtf12d on marin_semi.12d == FOWT.readQTF (raft_fowt.py:2081-2128) run live: bit for bit."""
    g = snapshot.load_fixture("f4_oc4semi_qtf12d.npz")
    heads, w, q = rq.read_qtf12d(os.path.join(REF, str(g["file"])), float(g["rho_water"]), float(g["g"]))
    assert np.array_equal(heads, g["heads_2nd"]) and np.array_equal(w, g["w1_2nd"])
    assert tuple(q.shape) == tuple(int(x) for x in g["qtf_shape"])
    iu = np.triu_indices(q.shape[0])
    assert np.array_equal(q[iu], g["qtf_upper"])
    # ... and the whole matrix (both triangles, the diagonal as printed) against readQTF run here and now
    import types
    from oracle import ref_harness as rh
    raft = rh.import_raft()
    holder = types.SimpleNamespace(rho_water=float(g["rho_water"]), g=float(g["g"]), nDOF=6)
    raft.raft_fowt.FOWT.readQTF(holder, os.path.join(REF, str(g["file"])))
    assert np.array_equal(q, holder.qtf) and np.array_equal(w, holder.w1_2nd) and np.array_equal(heads, holder.heads_2nd)
    off = ~np.eye(q.shape[0], dtype=bool)
    assert np.array_equal(q[off], np.conj(np.swapaxes(q, 0, 1))[off])          # Hermitian fill off the diagonal (:2125-2128)


def external_qtf_model():
    """The stand-in model of the f4 fixture with the parsed QTF attached (what FOWT.__init__ leaves behind when
    potSecOrder == 2, raft_fowt.py:427-431)."""
    fx, model = load_model_fixture("f4_oc4semi_qtf12d.npz")
    f = model.fowtList[0]
    shape = tuple(int(x) for x in fx["qtf_shape"])
    q = np.zeros(shape, dtype=complex)
    iu = np.triu_indices(shape[0])
    q[iu] = fx["qtf_upper"]
    ql = np.conj(np.swapaxes(q, 0, 1))
    il = np.tril_indices(shape[0], -1)
    q[il] = ql[il]
    f.qtf, f.heads_2nd, f.w1_2nd = q, list(fx["heads_2nd"]), np.array(fx["w1_2nd"])
    assert f.potSecOrder == 2
    return fx, model


def check_external_qtf_solve(ctx, tol, qtf_backend=None):
    from raft_amd import dropin
    fx, model = external_qtf_model()
    eng = dropin.Engine(ctx, qtf_backend=qtf_backend)
    for c in fx["cases"]:
        Xi = eng.solveDynamics(model, case_from_fixture(c))
        u, f = c["units"][0], model.fowtList[0]
        assert int(model._raftx_niter[0]) == int(u["niter"])
        assert rel_err(f.Fhydro_2nd, u["Fhydro_2nd"]) < tol and rel_err(f.Fhydro_2nd_mean, u["Fhydro_2nd_mean"]) < tol
        nH = Xi.shape[0] - 1
        assert group_rel_err(Xi[:nH], c["Xi"][:nH]) < tol
        assert rel_err(f.Z, u["Z"]) < tol
        assert np.any(np.abs(u["Fhydro_2nd"]) > 0)


def test_oracle_external_qtf_solveDynamics(oracle_ctx):
    """potSecOrder == 2 (raft_model.py:1037-1038): the force spectrum of the external QTF joins the excitation of the
    drag fixed point -- oracle backend, host force spectrum, against the live reference."""
    check_external_qtf_solve(oracle_ctx, 1e-9, qtf_backend=lambda *a: None)


@pytest.mark.gpu
def test_hip_external_qtf_solveDynamics(hip_ctx):
    """The same on the device: heading selection on the host, bilinear interpolation + diagonal sums of the QTF in
    raftx_qtf_force, fused fixed point with the second-order force as F_extra."""
    check_external_qtf_solve(hip_ctx, 1e-9)
