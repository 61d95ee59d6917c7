"""CPU suite for the second-order slender-body QTF: pins oracle/qtf_oracle.py +
raft_amd.qtf (packer, Kim & Yue feeder, force spectrum) against the reference's own goldens
(tests/test_fowt.py:192-216 pickles, fixed body) and against live-reference QTFs with body motions."""
import numpy as np
import pytest

from oracle import qtf_oracle
from raft_amd import qtf as rq
from raft_amd import snapshot as standin
from tests.util import rel_err

NAMES = ["VolturnUS-S", "VolturnUS-S-pointInertia"]


def _setup(name):
    fx = standin.load_fixture("refgold_qtf_%s.npz" % name)
    model = standin.build_model(fx["model"])
    f = model.fowtList[0]
    return fx, f, rq.pack_qtf(f)


@pytest.mark.parametrize("name", NAMES)
def test_reference_golden_qtf_fixed_body(name):
    fx, f, tab = _setup(name)
    w2, k2 = f.w1_2nd, f.k1_2nd
    kay = rq.kay_correction(tab.kay_geom, w2, k2, fx["fixed_beta"], f.depth, rho=f.rho_water, g=f.g)
    q = qtf_oracle.qtf_slender_body(tab, np.zeros((6, len(w2))), fx["fixed_beta"], w2, k2, f.depth, f.rho_water, f.g,
                                    f.M_struc, kay)
    np.testing.assert_allclose(q, fx["fixed_qtf"], rtol=1e-5, atol=1e-3)      # the reference's own gate
    assert rel_err(q, fx["fixed_qtf"]) < 1e-10


@pytest.mark.parametrize("name", NAMES)
def test_live_reference_qtf_with_motions(name):
    fx, f, tab = _setup(name)
    w2, k2 = f.w1_2nd, f.k1_2nd
    beta = fx["motion_beta"]
    kay = rq.kay_correction(tab.kay_geom, w2, k2, beta, f.depth, rho=f.rho_water, g=f.g)
    q = qtf_oracle.qtf_slender_body(tab, fx["motion_Xi2"], beta, w2, k2, f.depth, f.rho_water, f.g, f.M_struc, kay)
    assert rel_err(q, fx["motion_qtf"]) < 1e-10
    f_mean, f2 = rq.hydro_force_2nd(q, w2, f.w, f.dw, fx["motion_S0"])
    assert rel_err(f_mean, fx["motion_f_mean"]) < 1e-9
    assert rel_err(f2, fx["motion_f2"]) < 1e-9


def _numpy_qtf_backend(tabs, Xi, beta, w2, k2, depth, rho, g, Ms, kay):
    return np.array([qtf_oracle.qtf_slender_body(t, Xi[i], beta[i], w2, k2, depth, rho, g, Ms[i], kay[i])
                     for i, t in enumerate(tabs)])


@pytest.mark.parametrize("fixture", ["c5_internal_qtf.npz", "c5_oc4semi_qtf.npz"])
def test_c5_internal_qtf_solveDynamics(oracle_ctx, fixture):
    """potSecOrder == 1 through the drop-in Model.solveDynamics: converge, QTFs from the converged motions,
    second-order force into F_lin, iterate again from the same Xi_last (raft_model.py:1108-1131) -- against the
    live reference (host logic + CPU oracle + numpy QTF oracle)."""
    from raft_amd import dropin
    from tests.util import load_model_fixture, case_from_fixture, group_rel_err
    fx, model = load_model_fixture(fixture)
    eng = dropin.Engine(oracle_ctx, qtf_backend=_numpy_qtf_backend)
    for c in fx["cases"]:
        Xi = eng.solveDynamics(model, case_from_fixture(c))
        u = c["units"][0]
        f = model.fowtList[0]
        assert int(model._raftx_niter[0]) == int(u["niter"])
        assert rel_err(f.qtf[:, :, 0, :], u["qtf"]) < 1e-9
        assert rel_err(f.Fhydro_2nd, u["Fhydro_2nd"]) < 1e-9
        assert rel_err(f.Fhydro_2nd_mean, u["Fhydro_2nd_mean"]) < 1e-9
        nH = Xi.shape[0] - 1
        assert group_rel_err(Xi[:nH], c["Xi"][:nH]) < 1e-9
        assert rel_err(f.Z, u["Z"]) < 1e-9


def test_c5_full_size_oc4semi_200x200(oracle_ctx):
    """BASELINE configs[4] at its real shape (OC4semi-RAFT_QTF, nw = 200, 200 x 200 second-order grid, 0 and 30 deg):
    the oracle chain (C restatement + numpy QTF restatement + host logic) against the live reference's potSecOrder == 1
    solveDynamics (tests/golden/c5_oc4semi_full.npz, 7 minutes of reference time per case)."""
    from raft_amd import dropin
    from tests.util import load_model_fixture, case_from_fixture, group_rel_err
    fx, model = load_model_fixture("c5_oc4semi_full.npz")
    eng = dropin.Engine(oracle_ctx, qtf_backend=_numpy_qtf_backend)
    iu = np.triu_indices(200)
    for c in fx["cases"]:
        Xi = eng.solveDynamics(model, case_from_fixture(c))
        u = c["units"][0]
        f = model.fowtList[0]
        assert int(model._raftx_niter[0]) == int(u["niter"])
        assert rel_err(f.qtf[:, :, 0, :][iu], u["qtf_triu"]) < 1e-9
        assert rel_err(f.Fhydro_2nd, u["Fhydro_2nd"]) < 1e-9
        assert rel_err(f.Fhydro_2nd_mean, u["Fhydro_2nd_mean"]) < 1e-9
        assert group_rel_err(Xi[:1], np.asarray(c["Xi"])[:1]) < 1e-9


def test_qtf_12d_file_round_trip(tmp_path):
    fx, f, tab = _setup("VolturnUS-S")
    q4 = fx["motion_qtf"][:, :, None, :]
    path = str(tmp_path / "q.12d")
    rq.write_qtf12d(path, q4, f.w1_2nd, [fx["motion_beta"] % (2 * np.pi)], f.rho_water, f.g)
    heads, w, q = rq.read_qtf12d(path, f.rho_water, f.g)
    assert len(w) == len(f.w1_2nd) and np.allclose(w, f.w1_2nd, rtol=1e-4)
    assert rel_err(q[:, :, 0, :], fx["motion_qtf"]) < 2e-4          # the file carries 5 significant digits


@pytest.mark.parametrize("deck", ["refgold_qtf_VolturnUS-S.npz", "c5_oc4semi_qtf.npz"])
def test_strip_frame_formulation_of_the_device_kernel(deck):
    """k_qtf_pairs evaluates the strip terms in each strip's own frame (tests/qtf_device_model.py states the formulation
    in numpy): it must equal the term-by-term restatement of the reference -- here on a 24-point grid with body motions,
    vertical columns, pontoons, inclined braces and heave plates (Ca_p1 != Ca_p2 members included)."""
    from raft_amd import waves
    from tests import qtf_device_model as dm
    fx = standin.load_fixture(deck)
    f = standin.build_model(fx["model"]).fowtList[0]
    tab = rq.pack_qtf(f)
    nw2 = 24
    w2 = np.arange(1, nw2 + 1) * 0.02 * 2 * np.pi
    k2 = np.array([waves.wave_number(x, f.depth) for x in w2])
    rng = np.random.default_rng(1)
    amp = np.array([1.0, 0.3, 0.7, 0.01, 0.02, 0.004])[:, None] / (1.0 + (w2[None, :] / 0.6) ** 2)
    Xi = amp * np.exp(1j * (rng.uniform(0, 6, 6)[:, None] + 1.5 * w2[None, :]))
    beta = 0.6
    full = qtf_oracle.qtf_slender_body(tab, Xi, beta, w2, k2, f.depth, f.rho_water, f.g, f.M_struc)
    rest = qtf_oracle.qtf_slender_body(rq.QtfTable(np.zeros((0, rq.QS_N)), tab.members, tab.kay_geom), Xi, beta, w2, k2, f.depth,
                                       f.rho_water, f.g, f.M_struc)
    want = np.transpose(full - rest, (2, 0, 1))                     # the strip terms alone
    with np.errstate(invalid="ignore"):
        got = dm.qtf_strips(tab, Xi, beta, w2, k2, f.depth, f.rho_water, f.g)
    iu = np.triu_indices(nw2)
    assert rel_err(got[:, iu[0], iu[1]], want[:, iu[0], iu[1]]) < 1e-12
