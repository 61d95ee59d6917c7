"""GPU suite for the slender-body QTF kernels (through the C-ABI) against the reference's own goldens, the
live-reference QTFs with motions, and the numpy oracle on synthetic batches."""
import numpy as np
import pytest

from oracle import qtf_oracle
from raft_amd import qtf as rq
from raft_amd import snapshot as standin
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


def test_kim_yue_table_on_device_matches_host_scipy(hip_ctx):
    """raftx_qtf_kay (device j0/y0/jn/yn Hankel sums) against raft_amd.qtf.kay_correction (SciPy hankel1), which the
    live-reference C5 goldens pin; OC4semi's MacCamy-Fuchs columns with heave plates, two headings, 60-point grid; then
    the QTF that consumes the resident table against the one fed with the host table."""
    from raft_amd import qtf as rq, waves
    fx = standin.load_fixture("c5_oc4semi_qtf.npz")
    f = standin.build_model(fx["model"]).fowtList[0]
    tab = rq.pack_qtf(f)
    assert len(tab.kay_geom) >= 3
    nw2 = 60
    w2 = np.arange(1, nw2 + 1) * 0.008 * 2 * np.pi
    k2 = np.array([waves.wave_number(x, f.depth) for x in w2])
    betas = np.array([0.0, np.deg2rad(30.0)])
    host = np.array([rq.kay_correction(tab.kay_geom, w2, k2, b, f.depth, rho=f.rho_water, g=f.g) for b in betas])
    dev = hip_ctx.qtf_kay([tab, tab], betas, w2, k2, f.depth, f.rho_water, f.g, fetch=True)
    assert rel_err(dev, host) < 1e-10
    assert not np.any(dev[:, np.tril_indices(nw2, -1)[0], np.tril_indices(nw2, -1)[1]])     # upper triangle only
    rng = np.random.default_rng(3)
    Xi = 0.2 * (rng.normal(size=(2, 6, nw2)) + 1j * rng.normal(size=(2, 6, nw2))) / (1 + w2[None, None, :] ** 2)
    Ms = np.array([f.M_struc, f.M_struc])
    a = hip_ctx.qtf_slender([tab, tab], Xi, betas, w2, k2, f.depth, f.rho_water, f.g, Ms, None)      # consumes the table
    b = hip_ctx.qtf_slender([tab, tab], Xi, betas, w2, k2, f.depth, f.rho_water, f.g, Ms, host)
    c = hip_ctx.qtf_slender([tab, tab], Xi, betas, w2, k2, f.depth, f.rho_water, f.g, Ms, None)      # one-shot: gone now
    assert rel_err(a, b) < 1e-10
    assert rel_err(c, b) > 1e-6


def test_qtf_interleaved_row_partition_sums_to_the_full_matrix(hip_ctx):
    """raftx_qtf_slender_rows: the rows rank::world of one 200-point QTF for world = 3; the partial matrices have
    disjoint support (rows + Hermitian mirrors) and add up, bit for bit, to the unpartitioned result."""
    fx, f, tab = _setup("VolturnUS-S")
    rng = np.random.default_rng(21)
    nw2 = 200
    w2 = np.arange(1, nw2 + 1) * 0.0025 * 2 * np.pi
    from raft_amd import waves
    k2 = np.array([waves.wave_number(x, f.depth) for x in w2])
    amp = np.array([1.0, 0.3, 0.7, 0.01, 0.02, 0.004])[:, None] / (1.0 + (w2[None, :] / 0.6) ** 2)
    Xi = (amp * np.exp(1j * (rng.uniform(0, 6, 6)[:, None] + 1.5 * w2[None, :])))[None]
    args = ([tab], Xi, [np.deg2rad(30.0)], w2, k2, f.depth, f.rho_water, f.g, np.array([f.M_struc]))
    full = hip_ctx.qtf_slender(*args)
    world = 3
    parts = [hip_ctx.qtf_slender(*args, rows=(r, world)) for r in range(world)]
    support = sum((np.abs(p) > 0).astype(int) for p in parts)
    assert support.max() == 1                                         # no entry computed twice
    total = parts[0] + parts[1] + parts[2]
    assert np.array_equal(total.view(np.float64), full.view(np.float64))
    rows0 = np.nonzero(np.any(np.abs(parts[0][0]) > 0, axis=(1, 2)))[0]
    assert set(range(0, nw2, world)) <= set(rows0)
    with pytest.raises(Exception):
        hip_ctx.qtf_slender(*args, rows=(3, 3))


@pytest.mark.parametrize("fixture", ["c5_internal_qtf.npz", "c5_oc4semi_qtf.npz"])
def test_c5_internal_qtf_solveDynamics(hip_ctx, fixture):
    """BASELINE configs[4] path end to end on the device: first-order fixed point, slender-body QTF kernels fed with
    the converged motions, second-order force, restarted fixed point (raft_model.py:1108-1131)."""
    from raft_amd import dropin
    from tests.util import load_model_fixture, case_from_fixture, group_rel_err
    fx, model = load_model_fixture(fixture)
    eng = dropin.Engine(hip_ctx)
    for c in fx["cases"]:
        Xi = eng.solveDynamics(model, case_from_fixture(c))
        u = c["units"][0]
        f = model.fowtList[0]
        assert int(model._raftx_niter[0]) == int(u["niter"])
        assert rel_err(f.qtf[:, :, 0, :], u["qtf"]) < TOL
        assert rel_err(f.Fhydro_2nd, u["Fhydro_2nd"]) < TOL
        nH = Xi.shape[0] - 1
        assert group_rel_err(Xi[:nH], c["Xi"][:nH]) < TOL
        assert rel_err(f.Z, u["Z"]) < TOL


def test_dynamic_mooring_with_internal_qtf(hip_ctx, oracle_ctx):
    """moorMod == 2 together with potSecOrder == 1 (raft_model.py:1069-1072 and :1108-1131 in one per-unit loop): the stepped
    fixed point with the QTF re-entry inside it, device (QTF kernels included) against the oracle chain through the same
    drop-in code -- same MoorPy stand-in call sequence, same iteration counts (the NumPy path is compared with it on live
    objects in tests/test_dropin_live_reference.py::test_installed_dynamic_mooring_with_internal_qtf)."""
    from raft_amd import dropin
    from tests.util import load_model_fixture, case_from_fixture, group_rel_err, attach_fake_lines
    from tests.test_qtf import _numpy_qtf_backend
    fx, m_gpu = load_model_fixture("c5_internal_qtf.npz")
    _, m_cpu = load_model_fixture("c5_internal_qtf.npz")
    for m in (m_gpu, m_cpu):
        m.nIter = 10
        attach_fake_lines(m)
    case = fx["cases"][0]
    Xi_gpu = dropin.Engine(hip_ctx).solveDynamics(m_gpu, case_from_fixture(case)).copy()
    Xi_cpu = dropin.Engine(oracle_ctx, qtf_backend=_numpy_qtf_backend).solveDynamics(m_cpu, case_from_fixture(case)).copy()
    fg, fc = m_gpu.fowtList[0], m_cpu.fowtList[0]
    assert fg.ms.calls == fc.ms.calls and fg.ms.calls >= 3
    assert np.array_equal(m_gpu._raftx_niter, m_cpu._raftx_niter)
    assert rel_err(fg.qtf, fc.qtf) < TOL
    assert rel_err(fg.Fhydro_2nd, fc.Fhydro_2nd) < TOL
    nH = Xi_gpu.shape[0] - 1
    assert group_rel_err(Xi_gpu[:nH], Xi_cpu[:nH]) < TOL
    assert rel_err(fg.Z, fc.Z) < TOL


def test_restart_from_linearisation_point(hip_ctx, oracle_ctx):
    """raftx_set/fetch_linearisation_point: a restarted solve from the exported Xi_last reproduces the converged
    response in one iteration, identically on both libraries."""
    from tests.util import random_strips, random_matrices, synthetic_cases
    rng = np.random.default_rng(31)
    tables = [random_strips(rng, S) for S in (30, 12)]
    M0, B0, C0, _ = random_matrices(rng, 2)
    w, k, zeta, beta = synthetic_cases(rng, 2, 1, 90)
    res = []
    for ctx in (hip_ctx, oracle_ctx):
        ctx.upload_designs(tables, M0, B0, C0, len(w))
        ctx.upload_cases(w, k, 200.0, 1025.0, 9.81, zeta, beta)
        ctx.set_linearisation_point(None, keep_last=True)
        a = ctx.solve_dynamics(8)
        xl = ctx.fetch_linearisation_point()
        ctx.set_linearisation_point(xl, keep_last=False)
        b = ctx.solve_dynamics(8)
        assert np.all(a["flags"] & 1) and np.all(b["niter"] == 1)
        res.append((a, b, xl))
    assert rel_err(res[0][2], res[1][2]) < TOL
    for d in range(2):
        from tests.util import group_rel_err
        assert group_rel_err(res[0][1]["Xi"][d], res[0][0]["Xi"][d]) < 1e-13
        assert group_rel_err(res[0][1]["Xi"][d], res[1][1]["Xi"][d]) < TOL


def test_qtf_force_kernel(hip_ctx):
    """raftx_qtf_force against the SciPy restatement of calcHydroForce_2ndOrd (pinned on the live reference in
    tests/test_qtf.py), from a host QTF and from the QTFs left resident by raftx_qtf_slender."""
    fx, f, tab = _setup("VolturnUS-S")
    w2, k2 = f.w1_2nd, f.k1_2nd
    beta = fx["motion_beta"]
    S0 = fx["motion_S0"]
    fm, ff = hip_ctx.qtf_force(w2, f.w, f.dw, np.array([S0, 0.5 * S0]), qtf=np.array([fx["motion_qtf"], fx["fixed_qtf"]]))
    assert rel_err(ff[0], fx["motion_f2"]) < TOL and rel_err(fm[0], fx["motion_f_mean"]) < TOL
    r_mean, r_f = rq.hydro_force_2nd(fx["fixed_qtf"], w2, f.w, f.dw, 0.5 * S0)
    assert rel_err(ff[1], r_f) < TOL and rel_err(fm[1], r_mean) < TOL
    # resident path: QTFs never leave the device
    kay = rq.kay_correction(tab.kay_geom, w2, k2, beta, f.depth, rho=f.rho_water, g=f.g)
    none = hip_ctx.qtf_slender([tab], fx["motion_Xi2"][None], [beta], w2, k2, f.depth, f.rho_water, f.g, f.M_struc[None],
                               kay[None], fetch=False)
    assert none is None
    fm2, ff2 = hip_ctx.qtf_force(w2, f.w, f.dw, S0[None], qtf=None, n_set=1)
    assert rel_err(ff2[0], fx["motion_f2"]) < TOL and rel_err(fm2[0], fx["motion_f_mean"]) < TOL


def test_second_order_sweep_matches_dropin(hip_ctx):
    """Sweep.run_second_order (batch: 2 copies of the design x 2 sea states) against the live-reference
    potSecOrder == 1 solveDynamics results of the same deck."""
    from raft_amd import dropin
    from tests.util import load_model_fixture, case_from_fixture, group_rel_err
    fx, model = load_model_fixture("c5_internal_qtf.npz")
    f = model.fowtList[0]
    cases = [case_from_fixture(c) for c in fx["cases"]]
    sweep = dropin.sweep_from_models([model, model], cases)
    tab = rq.pack_qtf(f)
    w2, k2 = f.w1_2nd, f.k1_2nd
    from raft_amd import waves
    S0 = np.array([waves.sea_state(dict(c), f.w, f.dw)[2][0] for c in cases])
    kay = [[rq.kay_correction(tab.kay_geom, w2, k2, sweep.beta[c, 0], f.depth, rho=f.rho_water, g=f.g) for c in range(2)]] * 2
    # host Kim & Yue tables, then everything on the device (RAOs from the resident responses, Kim & Yue tables, QTFs,
    # force spectra): both against the live reference
    for kw in (dict(kay=kay), dict()):
        out = sweep.run_second_order(hip_ctx, [tab, tab], np.array([f.M_struc, f.M_struc]), w2, k2, S0,
                                     rho_water=f.rho_water, **kw)
        for d in range(2):
            for i, c in enumerate(fx["cases"]):
                u = c["units"][0]
                assert int(out["niter"][d, i]) == int(u["niter"])
                assert rel_err(out["Fhydro_2nd"][d, i], u["Fhydro_2nd"][0].real) < TOL
                assert group_rel_err(out["Xi"][d, i, :1], c["Xi"][:1]) < TOL


def test_resident_rao_path_equals_host_interpolation(hip_ctx):
    """raftx_qtf_slender with Xi == NULL: RAOs of the resident responses, interpolated on the device, against the same
    QTF fed with np.interp'ed RAOs (grid points inside, on and outside the first-order grid; a zero-amplitude bin)."""
    from raft_amd import dropin, waves
    from tests.util import load_model_fixture, case_from_fixture
    fx, model = load_model_fixture("c5_internal_qtf.npz")
    f = model.fowtList[0]
    cases = [case_from_fixture(c) for c in fx["cases"]]
    sweep = dropin.sweep_from_models([model], cases)
    sweep.zeta = sweep.zeta.copy()
    sweep.zeta[:, :, 7] = 0.0                                    # RAO := 0 where |zeta| <= 1e-6
    sweep.solve(hip_ctx)
    r = hip_ctx.fetch_results(want_Xi=True)
    tab = rq.pack_qtf(f)
    w2 = np.concatenate([[0.5 * f.w[0]], f.w[3:9], np.linspace(f.w[10], f.w[-1] * 1.2, 25)])      # below, on, between, above
    k2 = np.array([waves.wave_number(x, f.depth) for x in w2])
    nC = len(cases)
    Xi2 = np.zeros((nC, 6, len(w2)), dtype=complex)
    for c in range(nC):
        rao = waves.get_rao(r["Xi"][0, c, 0], sweep.zeta[c, 0])
        for j in range(6):
            Xi2[c, j] = np.interp(w2, f.w, rao[j], left=0, right=0)
    args = ([tab] * nC, sweep.beta[:, 0], w2, k2, f.depth, f.rho_water, f.g, np.array([f.M_struc] * nC))
    a = hip_ctx.qtf_slender(args[0], None, *args[1:])
    b = hip_ctx.qtf_slender(args[0], Xi2, *args[1:])
    assert rel_err(a, b) < 1e-12
    with pytest.raises(Exception):
        hip_ctx.qtf_slender(args[0][:1], None, sweep.beta[:1, 0], w2, k2, f.depth, f.rho_water, f.g, np.array([f.M_struc]))
