"""GPU suite for the slender-body QTF kernels (through the C-ABI) against the reference's own goldens, the
live-reference QTFs with motions, and the numpy oracle on synthetic batches."""
import numpy as np
import pytest

from oracle import qtf_oracle
from raft_amd import qtf as rq
from tests import standin
from tests.util import rel_err

pytestmark = pytest.mark.gpu
NAMES = ["VolturnUS-S", "VolturnUS-S-pointInertia"]
TOL = 1e-9


def _setup(name):
    fx = standin.load_fixture("refgold_qtf_%s.npz" % name)
    model = standin.build_model(fx["model"])
    f = model.fowtList[0]
    return fx, f, rq.pack_qtf(f)


@pytest.mark.parametrize("name", NAMES)
def test_reference_golden_qtf_fixed_body(name, hip_ctx):
    fx, f, tab = _setup(name)
    w2, k2 = f.w1_2nd, f.k1_2nd
    kay = rq.kay_correction(tab.kay_geom, w2, k2, fx["fixed_beta"], f.depth, rho=f.rho_water, g=f.g)
    q = hip_ctx.qtf_slender([tab], np.zeros((1, 6, len(w2))), [fx["fixed_beta"]], w2, k2, f.depth, f.rho_water, f.g,
                            f.M_struc[None], kay[None])[0]
    np.testing.assert_allclose(q, fx["fixed_qtf"], rtol=1e-5, atol=1e-3)      # the reference's own gate
    assert rel_err(q, fx["fixed_qtf"]) < TOL


@pytest.mark.parametrize("name", NAMES)
def test_live_reference_qtf_with_motions(name, hip_ctx):
    fx, f, tab = _setup(name)
    w2, k2 = f.w1_2nd, f.k1_2nd
    beta = fx["motion_beta"]
    kay = rq.kay_correction(tab.kay_geom, w2, k2, beta, f.depth, rho=f.rho_water, g=f.g)
    q = hip_ctx.qtf_slender([tab], fx["motion_Xi2"][None], [beta], w2, k2, f.depth, f.rho_water, f.g, f.M_struc[None],
                            kay[None])[0]
    assert rel_err(q, fx["motion_qtf"]) < TOL
    f_mean, f2 = rq.hydro_force_2nd(q, w2, f.w, f.dw, fx["motion_S0"])
    assert rel_err(f2, fx["motion_f2"]) < TOL


def test_qtf_batch_against_numpy_oracle(hip_ctx):
    """BASELINE configs[4] shape: a 200-bin second-order grid, two sets (fixed body at 0 deg, moving body at
    30 deg) in one batch; size-independent property: the result is Hermitian in (w1, w2)."""
    fx, f, tab = _setup("VolturnUS-S")
    rng = np.random.default_rng(12)
    nw2 = 200
    w2 = np.arange(1, nw2 + 1) * 0.0025 * 2 * np.pi
    from raft_amd import waves
    k2 = np.array([waves.wave_number(x, f.depth) for x in w2])
    Xi = np.zeros((2, 6, nw2), dtype=complex)
    amp = np.array([1.0, 0.3, 0.7, 0.01, 0.02, 0.004])[:, None] / (1.0 + (w2[None, :] / 0.6) ** 2)
    Xi[1] = amp * np.exp(1j * (rng.uniform(0, 6, 6)[:, None] + 1.5 * w2[None, :]))
    betas = [0.0, np.deg2rad(30.0)]
    q = hip_ctx.qtf_slender([tab, tab], Xi, betas, w2, k2, f.depth, f.rho_water, f.g, np.array([f.M_struc, f.M_struc]))
    for s in range(2):
        ref = qtf_oracle.qtf_slender_body(tab, Xi[s], betas[s], w2, k2, f.depth, f.rho_water, f.g, f.M_struc)
        assert rel_err(q[s], ref) < TOL
        off = ~np.eye(nw2, dtype=bool)                                  # the diagonal is left as computed (raft_fowt.py:2070)
        assert np.array_equal(q[s][off], np.conj(np.transpose(q[s], (1, 0, 2)))[off])
