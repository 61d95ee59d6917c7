"""CPU suite, only where /root/reference is present (skipped on the GPU box): the drop-in methods installed INTO the
live, unmodified reference package (raft_amd.dropin.install) and driven through the reference's own objects --
Model / FOWT / Member instances, not the stand-ins -- with the CPU oracle as the backend.  Proves that the binding
of INTEGRATION.md works on the real attribute surface, and that results equal the reference's NumPy path (computed
with the package un-patched)."""
import contextlib
import copy
import os

import numpy as np
import pytest

from oracle import ref_harness as rh
from tests.util import group_rel_err, rel_err, attach_fake_lines

pytestmark = pytest.mark.skipif(not rh.tree_available(), reason="reference tree not present (these tests read its decks)")


def _numpy_qtf_backend(tabs, Xi, beta, w2, k2, depth, rho, g, Ms, kay):
    from oracle import qtf_oracle
    return np.array([qtf_oracle.qtf_slender_body(t, Xi[i], beta[i], w2, k2, depth, rho, g, Ms[i], kay[i])
                     for i, t in enumerate(tabs)])


class Patch:
    def __init__(self, oracle_ctx):
        from raft_amd import dropin
        rh.import_raft()
        self.dropin = dropin
        dropin._default_engine = dropin.Engine(oracle_ctx, qtf_backend=_numpy_qtf_backend)
        self.saved = dropin.install()

    @contextlib.contextmanager
    def unpatched(self):
        """The reference's own NumPy methods, temporarily restored."""
        self.dropin.uninstall(self.saved)
        try:
            yield
        finally:
            self.dropin.install()

    def close(self):
        self.dropin.uninstall(self.saved)
        self.dropin._default_engine = self.dropin.Engine()


@pytest.fixture()
def patch(oracle_ctx):
    p = Patch(oracle_ctx)
    yield p
    p.close()


def _model(deck, settings):
    d = rh.prepare_design(rh.load_design(os.path.join(rh.REFERENCE_ROOT, deck)), settings=settings)
    d["platform"].pop("outFolderQTF", None)
    m = rh.build_model(d)
    for f in m.fowtList:
        f.outFolderQTF = None
    return m


def test_installed_solveDynamics_equals_numpy_path(patch):
    """OC3spar (C1 settings): the patched Model.solveDynamics on live objects vs the reference's NumPy path."""
    from raft import raft_model
    settings = dict(min_freq=0.008, max_freq=0.4, nIter=10, XiStart=0)
    case = rh.make_case(Hs=2.0, Tp=8.0, heading=20.0)
    m_new, m_old = _model("designs/OC3spar.yaml", settings), _model("designs/OC3spar.yaml", settings)
    assert raft_model.Model.solveDynamics is patch.dropin.solveDynamics
    Xi_new = m_new.solveDynamics(copy.deepcopy(case)).copy()
    with patch.unpatched():
        assert raft_model.Model.solveDynamics is not patch.dropin.solveDynamics
        Xi_old = m_old.solveDynamics(copy.deepcopy(case)).copy()
    assert Xi_new.shape == Xi_old.shape
    assert group_rel_err(Xi_new[:1], Xi_old[:1]) < 1e-10
    fn, fo = m_new.fowtList[0], m_old.fowtList[0]
    assert rel_err(fn.Z, fo.Z) < 1e-12
    assert rel_err(fn.B_hydro_drag, fo.B_hydro_drag) < 1e-10
    assert rel_err(fn.Xi_fullDOF, fo.Xi_fullDOF) < 1e-10
    assert m_new.results['response'] == {}


def test_installed_fowt_methods_on_live_objects(patch):
    """FOWT.calcHydroExcitation / calcHydroLinearization / calcDragExcitation patched into the live package,
    against the reference's NumPy methods on a twin model (VolturnUS-S test deck: MacCamy-Fuchs columns)."""
    settings = dict(nIter=4)
    case = {'wave_spectrum': 'JONSWAP', 'wave_heading': [10, -35], 'wave_period': [9, 13], 'wave_height': [3, 5]}
    m_new = _model("tests/test_data/VolturnUS-S.yaml", settings)
    m_old = _model("tests/test_data/VolturnUS-S.yaml", settings)
    fn, fo = m_new.fowtList[0], m_old.fowtList[0]
    Xi = 0.2 * np.exp(1j * np.linspace(0, 5, 6 * fn.nw).reshape(6, fn.nw))
    fn.calcHydroExcitation(copy.deepcopy(case), memberList=fn.memberList)
    Bn = fn.calcHydroLinearization(Xi)
    Fn = [fn.calcDragExcitation(ih).copy() for ih in (0, 1)]
    with patch.unpatched():
        fo.calcHydroExcitation(copy.deepcopy(case), memberList=fo.memberList)
        Bo = fo.calcHydroLinearization(Xi)
        Fo = [fo.calcDragExcitation(ih).copy() for ih in (0, 1)]
    assert rel_err(fn.F_hydro_iner, fo.F_hydro_iner) < 1e-12
    assert np.array_equal(fn.zeta, fo.zeta) and np.array_equal(fn.beta, fo.beta)
    assert rel_err(Bn, Bo) < 1e-10
    for a, b in zip(Fn, Fo):
        assert rel_err(a, b) < 1e-10


def test_installed_internal_qtf_on_live_objects(patch):
    """potSecOrder == 1 through the patched live package (QTF + second-order force + restarted drag iteration)."""
    settings = dict(nIter=10)
    case = rh.make_case(Hs=5.0, Tp=11.0, heading=15.0)
    m_new = _model("tests/test_data/VolturnUS-S.yaml", settings)
    m_old = _model("tests/test_data/VolturnUS-S.yaml", settings)
    Xi_new = m_new.solveDynamics(copy.deepcopy(case)).copy()
    with patch.unpatched():
        Xi_old = m_old.solveDynamics(copy.deepcopy(case)).copy()
    assert group_rel_err(Xi_new[:1], Xi_old[:1]) < 1e-9
    assert rel_err(m_new.fowtList[0].qtf, m_old.fowtList[0].qtf) < 1e-9
    assert rel_err(m_new.fowtList[0].Fhydro_2nd, m_old.fowtList[0].Fhydro_2nd) < 1e-9


def test_member_description_sweep_equals_reference_models(oracle_ctx):
    """Design candidates handed over as member descriptions (no Model per candidate) against the live reference's
    own Model(...) -> calcStatics -> calcHydroConstants -> solveDynamics of every candidate."""
    import io
    from raft_amd import dropin, geometry as G
    from oracle.make_golden import volturnus_variant
    base = rh.prepare_design(rh.load_design(os.path.join(rh.REFERENCE_ROOT, "examples/VolturnUS-S_example.yaml")),
                             settings=dict(min_freq=0.01, max_freq=0.3))          # nw = 30 keeps the reference quick
    case = rh.make_case(Hs=4.0, Tp=9.0, heading=20.0)
    scales = np.array([[1.1, 0.9, 1.05, 0.95, 1.2], [0.8, 1.2, 0.9, 1.1, 0.85]])
    designs = [volturnus_variant(base, s) for s in scales]
    with contextlib.redirect_stdout(io.StringIO()):
        m0 = rh.build_model(copy.deepcopy(base))
        refs = []
        for d in designs:
            m = rh.build_model(copy.deepcopy(d))
            refs.append(m.solveDynamics(copy.deepcopy(case)).copy())
    tabs = G.concat_units([G.describe_unit(d) for d in designs])
    sweep = dropin.sweep_from_member_tables(m0, G.describe_unit(base), tabs, [case], oracle_ctx)
    out = sweep.run(oracle_ctx)
    for j, Xi_ref in enumerate(refs):
        assert group_rel_err(out["Xi"][j, 0, :1], Xi_ref[:1]) < 1e-9


@pytest.mark.parametrize("headings", [1, 2])
def test_saveTurbineOutputs_statistics_equal_reference(oracle_ctx, headings):
    """FOWT.saveTurbineOutputs (raft_fowt.py:2291-2745) through the statistics entry point: every key the
    reference writes for a rigid, MoorPy-less unit -- motions, nacelle accelerations, tower-base moment, wave PSD --
    against the reference's own method run on the reference's own responses."""
    import io
    from raft_amd import dropin
    with contextlib.redirect_stdout(io.StringIO()):
        m = _model("examples/VolturnUS-S_example.yaml", dict(min_freq=0.01, max_freq=0.3))
        m2 = _model("examples/VolturnUS-S_example.yaml", dict(min_freq=0.01, max_freq=0.3))
    case = rh.make_case(Hs=4.0, Tp=9.0, heading=20.0)
    if headings == 2:
        case.update(wave_heading=[20.0, -60.0], wave_spectrum=["JONSWAP", "JONSWAP"], wave_period=[9.0, 13.0],
                    wave_height=[4.0, 2.0], wave_gamma=[0, 0])
    with contextlib.redirect_stdout(io.StringIO()):
        m.solveDynamics(copy.deepcopy(case))
        ref = {}
        m.fowtList[0].saveTurbineOutputs(ref, copy.deepcopy(case))
    eng = dropin.Engine(oracle_ctx)
    eng.solveDynamics(m2, copy.deepcopy(case))
    got = eng.saveTurbineOutputs(m2.fowtList[0], {}, copy.deepcopy(case))
    assert set(got) <= set(ref)
    checked = 0
    for key, val in got.items():
        a, b = np.asarray(val), np.asarray(ref[key])
        assert a.shape == b.shape, key
        scale = max(np.max(np.abs(b)), 1e-300)
        assert np.max(np.abs(a - b)) <= 1e-8 * scale + 1e-12, (key, np.max(np.abs(a - b)) / scale)
        checked += 1
    assert checked >= 60 and got["Mbase_std"][0] > 1e6 and got["AxRNA_std"][0] > 0
    missing = set(ref) - set(got)
    assert all(k.startswith(("Tmoor", "wind_PSD", "cavitation")) for k in missing), missing
    with pytest.raises(dropin.UnsupportedFOWT):
        dropin.Engine(oracle_ctx).saveTurbineOutputs(m2.fowtList[0], {}, case)      # nothing resident for that engine
    if headings == 2:
        # VERDICT r5 missing 7, the quasi-static half: the mooring-tension block (raft_fowt.py:2356-2399, moorMod == 0) -- tension
        # amplitudes J Xi per heading and bin, their getRMS / getPSD (with the reference's own ``w[0]`` in place of dw), mean
        # tensions and +-3 sigma bounds -- as channels of the same statistics launch.  MoorPy is absent here: both sides ask the
        # same stand-in system for the Jacobian and the mean tensions.  Line dynamics (moorMod != 0) still raises.
        from tests.util import FakeStaticLines
        with _identity_lines2ss():
            m.fowtList[0].ms, m2.fowtList[0].ms = FakeStaticLines(), FakeStaticLines()
            with contextlib.redirect_stdout(io.StringIO()):
                ref2 = {}
                m.fowtList[0].saveTurbineOutputs(ref2, copy.deepcopy(case))
            got2 = eng.saveTurbineOutputs(m2.fowtList[0], {}, copy.deepcopy(case))
            for key in ("Tmoor_avg", "Tmoor_std", "Tmoor_max", "Tmoor_min", "Tmoor_PSD"):
                a, b = np.asarray(got2[key]), np.asarray(ref2[key])
                assert a.shape == b.shape and np.max(np.abs(a - b)) <= 1e-9 * np.max(np.abs(b)), key
            assert got2["Tmoor_std"].shape == (6,) and np.all(got2["Tmoor_std"] > 1e3) and got2["Tmoor_PSD"].shape == (6, m2.fowtList[0].nw)
            for key, val in got.items():                                          # the other blocks beside them: unchanged
                assert np.array_equal(np.asarray(val), np.asarray(got2[key])), key
            m2.fowtList[0].moorMod = 2
            with pytest.raises(dropin.UnsupportedFOWT, match="line-dynamics"):
                eng.saveTurbineOutputs(m2.fowtList[0], {}, copy.deepcopy(case))
            m2.fowtList[0].moorMod = 0


@contextlib.contextmanager
def _identity_lines2ss():
    """MoorPy's composite-line conversion as the identity, in the live reference module and in the harness's stand-in
    ``moorpy.helpers`` (whose functions raise): what both sides of a test with a stand-in line system call."""
    import sys
    rh.import_raft()
    import raft.raft_fowt as rf
    mph = sys.modules["moorpy.helpers"]
    saved = rf.lines2ss, mph.lines2ss
    rf.lines2ss = mph.lines2ss = lambda ms: ms
    try:
        yield
    finally:
        rf.lines2ss, mph.lines2ss = saved


def test_saveTurbineOutputs_of_the_flexible_deck(oracle_ctx):
    """VERDICT r5 missing 3: FOWT.saveTurbineOutputs for a unit with flexible members -- PRP motions from the rigid-body node,
    hub accelerations and the tower-base loads from the finite-element stiffness of the FLEXIBLE tower
    (raft_fowt.py:2299-2355, 2422-2444, 2540-2601) -- as linear channels of the reduced response through
    raftx_response_stats, against the reference's own method on the reference's own solve."""
    import io
    from raft_amd import dropin
    settings = dict(min_freq=0.01, max_freq=0.3, nIter=8, XiStart=0.1)
    with contextlib.redirect_stdout(io.StringIO()):
        m = _model("tests/test_data/VolturnUS-S-flexible.yaml", settings)
        m2 = _model("tests/test_data/VolturnUS-S-flexible.yaml", settings)
    for mm in (m, m2):
        for f in mm.fowtList:
            f.potSecOrder = 0
            cm = np.zeros((f.nDOF, f.nDOF))
            cm[:6, :6] = rh.DEFAULT_C_MOOR
            f.C_moor = cm
    assert m.fowtList[0].nDOF > 6 and m.fowtList[0].memberList[m.fowtList[0].nplatmems].type != "rigid"
    case = rh.make_case(Hs=[5.0, 2.5], Tp=[11.0, 8.0], heading=[25.0, -40.0], spectrum=["JONSWAP"] * 2, gamma=[0, 0])
    from tests.util import FakeStaticLines
    with contextlib.redirect_stdout(io.StringIO()):
        m.solveDynamics(copy.deepcopy(case))
        ref = {}
        m.fowtList[0].saveTurbineOutputs(ref, copy.deepcopy(case))
    eng = dropin.Engine(oracle_ctx)
    eng.solveDynamics(m2, copy.deepcopy(case))
    got = eng.saveTurbineOutputs(m2.fowtList[0], {}, copy.deepcopy(case))
    assert set(got) <= set(ref)
    checked = 0
    for key, val in got.items():
        a, b = np.asarray(val), np.asarray(ref[key])
        assert a.shape == b.shape, key
        scale = max(np.max(np.abs(b)), 1e-300)
        assert np.max(np.abs(a - b)) <= 1e-7 * scale + 1e-9, (key, np.max(np.abs(a - b)) / scale)
        checked += 1
    assert checked >= 90 and got["MbaseY_std"][0] > 1e5 and got["FbaseX_std"][0] > 1e3 and got["AxRNA_std"][0] > 0
    assert np.array_equal(got["Mbase_std"], got["MbaseY_std"])
    missing = set(ref) - set(got)
    assert all(k.startswith(("Tmoor", "wind_PSD", "cavitation")) for k in missing), missing
    # ... and with a quasi-static line system on the unit: the tension block (:2356-2399) over the PRP motions of the rigid-body node
    with _identity_lines2ss():
        m.fowtList[0].ms = FakeStaticLines(seed=9)
        m2.fowtList[0].ms = FakeStaticLines(seed=9)
        with contextlib.redirect_stdout(io.StringIO()):
            ref2 = {}
            m.fowtList[0].saveTurbineOutputs(ref2, copy.deepcopy(case))
        got2 = eng.saveTurbineOutputs(m2.fowtList[0], {}, copy.deepcopy(case))
    for key in ("Tmoor_avg", "Tmoor_std", "Tmoor_max", "Tmoor_min", "Tmoor_PSD"):
        a, b = np.asarray(got2[key]), np.asarray(ref2[key])
        assert a.shape == b.shape and np.max(np.abs(a - b)) <= 1e-7 * np.max(np.abs(b)), key
    assert np.all(got2["Tmoor_std"] > 1e3)
    m.fowtList[0].ms = m2.fowtList[0].ms = None
    with pytest.raises(dropin.UnsupportedFOWT):
        dropin.Engine(oracle_ctx).saveTurbineOutputs(m2.fowtList[0], {}, case)      # an engine that has not solved this unit


@pytest.mark.parametrize("nIter", [10, 2])
def test_installed_dynamic_mooring_hook(patch, nIter):
    """moorMod == 2 (raft_model.py:1022-1030,1069-1072): the mooring damping is re-evaluated on the host about every
    iterate; the patched solveDynamics steps the device fixed point one launch per iteration and must make the same
    calls, in the same order, with the same arguments as the NumPy path (a motion-dependent stand-in for MoorPy)."""
    settings = dict(min_freq=0.008, max_freq=0.4, nIter=nIter, XiStart=0.1)
    case = rh.make_case(Hs=4.0, Tp=9.0, heading=-25.0)
    m_new, m_old = _model("designs/OC3spar.yaml", settings), _model("designs/OC3spar.yaml", settings)
    attach_fake_lines(m_new)
    attach_fake_lines(m_old)
    Xi_new = m_new.solveDynamics(copy.deepcopy(case)).copy()
    with patch.unpatched():
        Xi_old = m_old.solveDynamics(copy.deepcopy(case)).copy()
    fn, fo = m_new.fowtList[0], m_old.fowtList[0]
    assert fn.ms.calls == fo.ms.calls and fn.ms.calls >= 3
    assert fn.ms.level == pytest.approx(fo.ms.level, rel=1e-10)
    assert group_rel_err(Xi_new[:1], Xi_old[:1]) < 1e-10
    assert rel_err(fn.Z, fo.Z) < 1e-10
    assert rel_err(fn.B_hydro_drag, fo.B_hydro_drag) < 1e-9
    assert rel_err(fn.Xi_fullDOF, fo.Xi_fullDOF) < 1e-10


@pytest.mark.parametrize("nIter", [10, 3])
def test_installed_dynamic_mooring_with_internal_qtf(patch, nIter):
    """moorMod == 2 AND potSecOrder == 1 in one solve (VERDICT r5 missing 2): upstream both live in the same per-unit loop --
    the mooring damping re-linearised about every iterate (raft_model.py:1069-1072) and, at the first convergence, the QTFs
    and second-order force computed from that response, the pass counter reset and the loop continued from the same
    linearisation point (:1108-1131).  The patched path steps the device fixed point and runs the re-entry inside the
    stepped loop: same MoorPy call sequence, same QTFs, same response."""
    settings = dict(nIter=nIter)
    case = rh.make_case(Hs=5.0, Tp=11.0, heading=15.0)
    m_new = _model("tests/test_data/VolturnUS-S.yaml", settings)
    m_old = _model("tests/test_data/VolturnUS-S.yaml", settings)
    assert m_new.fowtList[0].potSecOrder == 1
    attach_fake_lines(m_new)
    attach_fake_lines(m_old)
    Xi_new = m_new.solveDynamics(copy.deepcopy(case)).copy()
    with patch.unpatched():
        Xi_old = m_old.solveDynamics(copy.deepcopy(case)).copy()
    fn, fo = m_new.fowtList[0], m_old.fowtList[0]
    assert fn.ms.calls == fo.ms.calls and fn.ms.calls >= 3
    assert fn.ms.level == pytest.approx(fo.ms.level, rel=1e-10)
    assert group_rel_err(Xi_new[:1], Xi_old[:1]) < 1e-9
    assert hasattr(fn, "qtf") == hasattr(fo, "qtf") == (nIter == 10)       # three passes never converge: no re-entry on either side
    if hasattr(fo, "qtf"):
        assert rel_err(fn.qtf, fo.qtf) < 1e-9
    assert rel_err(fn.Fhydro_2nd, fo.Fhydro_2nd) < 1e-9
    assert rel_err(fn.Z, fo.Z) < 1e-9
    assert rel_err(fn.B_hydro_drag, fo.B_hydro_drag) < 1e-9


@pytest.mark.parametrize("unit_lines", [False, True])
def test_installed_array_level_dynamic_mooring(patch, unit_lines):
    """Array-level moorMod == 2 on the live farm deck (raft_model.py:1173-1182): the patched solveDynamics hands
    Model.updateMooringDynamicMatrices the same per-unit motions as the NumPy path (:1156,1178) and adds the same
    -w^2 (M + A) + i w B + C to the coupled system (a motion-dependent stand-in for the shared MoorPy system)."""
    from tests.util import attach_fake_array_lines
    settings = dict(min_freq=0.01, max_freq=0.3, nIter=6, XiStart=0.1)
    case = rh.make_case(Hs=5.0, Tp=10.0, heading=20.0)
    m_new, m_old = _model("designs/VolturnUS-S_farm.yaml", settings), _model("designs/VolturnUS-S_farm.yaml", settings)
    assert len(m_new.fowtList) > 1
    for m in (m_new, m_old):
        if unit_lines:
            attach_fake_lines(m)
        attach_fake_array_lines(m)
    Xi_new = m_new.solveDynamics(copy.deepcopy(case)).copy()
    with patch.unpatched():
        Xi_old = m_old.solveDynamics(copy.deepcopy(case)).copy()
    assert m_new.ms.calls == m_old.ms.calls == 1
    assert all(rel_err(a, b) < 1e-10 for a, b in zip(m_new.ms.seen, m_old.ms.seen))
    assert m_new.ms.level == pytest.approx(m_old.ms.level, rel=1e-10)
    assert group_rel_err(Xi_new[:1], Xi_old[:1]) < 1e-10
    for fn, fo in zip(m_new.fowtList, m_old.fowtList):
        assert rel_err(fn.Z, fo.Z) < 1e-10 and rel_err(fn.Xi_fullDOF, fo.Xi_fullDOF) < 1e-10


@pytest.mark.parametrize("deck", ["examples/VolturnUS-S_example.yaml", "tests/test_data/VolturnUS-S.yaml"])
def test_materialised_member_side_effects(patch, deck):
    """install(materialise_members=True): the per-member arrays the reference's methods leave behind (SURVEY.md 8b --
    mem.u, ud, pDyn raft_member.py:1927-1937; mem.F_hydro_iner :1991; mem.Bmat, mem.F_exc_drag :2117,2122) are filled
    from the library's per-strip export, equal to the NumPy path's, and the UN-REPLACED Member.calcDragExcitation
    (:2128-2152) run on them after the drop-in's methods gives what the reference's own chain gives."""
    settings = dict(min_freq=0.01, max_freq=0.3, nIter=6, XiStart=0.1)
    case = rh.make_case(Hs=[5.0, 2.0], Tp=[11.0, 8.0], heading=[20.0, -50.0], spectrum=["JONSWAP"] * 2, gamma=[0, 0])
    m_new, m_old = _model(deck, settings), _model(deck, settings)
    for f in (m_new.fowtList[0], m_old.fowtList[0]):
        f.potSecOrder = 0
    patch.dropin.install(materialise_members=True)
    try:
        fn, fo = m_new.fowtList[0], m_old.fowtList[0]
        fn.calcHydroExcitation(copy.deepcopy(case), memberList=fn.memberList)
        with patch.unpatched():
            fo.calcHydroExcitation(copy.deepcopy(case), memberList=fo.memberList)
        for a, b in zip(fn.memberList, fo.memberList):
            assert a.u.shape == b.u.shape and a.pDyn.shape == b.pDyn.shape
            assert rel_err(a.u, b.u) < 1e-12 and rel_err(a.ud, b.ud) < 1e-12 and rel_err(a.pDyn, b.pDyn) < 1e-12
            assert rel_err(a.F_hydro_iner, b.F_hydro_iner) < 1e-11
        Xi = 0.05 * np.exp(1j * np.linspace(0, 2 * np.pi, 6 * fn.nw)).reshape(6, fn.nw)
        B_new = fn.calcHydroLinearization(Xi)
        with patch.unpatched():
            B_old = fo.calcHydroLinearization(Xi)
        assert rel_err(B_new, B_old) < 1e-10
        for a, b in zip(fn.memberList, fo.memberList):
            assert rel_err(a.Bmat, b.Bmat) < 1e-10 and rel_err(a.F_exc_drag, b.F_exc_drag) < 1e-10
        # the reference's own Member.calcDragExcitation, un-replaced, on the members the drop-in has just written
        for ih in (1, 0):
            with patch.unpatched():
                F_ref = fo.calcDragExcitation(ih)
            assert rel_err(fn.calcDragExcitation(ih), F_ref) < 1e-10
            got = [a.calcDragExcitation(ih) for a in fn.memberList]
            want = [b.calcDragExcitation(ih) for b in fo.memberList]
            assert all(rel_err(g, w_) < 1e-10 for g, w_ in zip(got, want) if np.any(w_))
            assert all(rel_err(a.F_exc_drag, b.F_exc_drag) < 1e-10 for a, b in zip(fn.memberList, fo.memberList))
        # ... and after a whole solveDynamics: Bmat of the last linearisation, F_exc_drag of the last heading
        Xi_new = m_new.solveDynamics(copy.deepcopy(case)).copy()
        with patch.unpatched():
            Xi_old = m_old.solveDynamics(copy.deepcopy(case)).copy()
        assert group_rel_err(Xi_new[:2], Xi_old[:2]) < 1e-10
        for a, b in zip(fn.memberList, fo.memberList):
            assert rel_err(a.u, b.u) < 1e-12 and rel_err(a.Bmat, b.Bmat) < 1e-9 and rel_err(a.F_exc_drag, b.F_exc_drag) < 1e-9
    finally:
        patch.dropin.install(materialise_members=False)


def test_installed_calcHydroExcitation_full_dof_and_member_list(patch):
    """FOWT.calcHydroExcitation on live objects: F_hydro_iner_fullDOF (the per-member vectors about each member's own
    node, raft_fowt.py:1853-1857) and F_BEM_fullDOF are set as upstream sets them, and the member list means what it
    means upstream: an empty (default) list contributes no strip-theory excitation."""
    settings = dict(min_freq=0.01, max_freq=0.3)
    case = rh.make_case(Hs=4.0, Tp=10.0, heading=35.0)
    m_new = _model("examples/VolturnUS-S_example.yaml", settings)
    m_old = _model("examples/VolturnUS-S_example.yaml", settings)
    fn, fo = m_new.fowtList[0], m_old.fowtList[0]
    fn.calcHydroExcitation(copy.deepcopy(case), memberList=fn.memberList)
    with patch.unpatched():
        fo.calcHydroExcitation(copy.deepcopy(case), memberList=fo.memberList)
    assert fn.F_hydro_iner_fullDOF.shape == fo.F_hydro_iner_fullDOF.shape
    assert rel_err(fn.F_hydro_iner_fullDOF, fo.F_hydro_iner_fullDOF) < 1e-12
    assert rel_err(fn.F_hydro_iner, fo.F_hydro_iner) < 1e-12
    assert fn.F_BEM_fullDOF.shape == fo.F_BEM_fullDOF.shape and not np.any(fn.F_BEM_fullDOF)
    # a sub-list of members, then the default (empty) list
    sub_n, sub_o = fn.memberList[:3], fo.memberList[:3]
    fn.calcHydroExcitation(copy.deepcopy(case), memberList=sub_n)
    with patch.unpatched():
        fo.calcHydroExcitation(copy.deepcopy(case), memberList=sub_o)
    assert rel_err(fn.F_hydro_iner, fo.F_hydro_iner) < 1e-12
    assert rel_err(fn.F_hydro_iner_fullDOF, fo.F_hydro_iner_fullDOF) < 1e-12
    fn.calcHydroExcitation(copy.deepcopy(case))
    with patch.unpatched():
        fo.calcHydroExcitation(copy.deepcopy(case))
    assert not np.any(fo.F_hydro_iner) and not np.any(fn.F_hydro_iner)
    assert fn.F_hydro_iner.shape == fo.F_hydro_iner.shape and not np.any(fn.F_hydro_iner_fullDOF)


# ---------------------------------------------------------------------------------------------------------------------
# every platform deck of the reference tree that the reference itself can build and solve here (MoorPy / CCBlade stubbed,
# mooring replaced by the injected stiffness of SURVEY 8d): rigid units, farms, internal-QTF decks, the flexible decks
def _platform_decks():
    if not rh.tree_available():
        return []
    import glob
    out = []
    for sub in ("designs", "examples", "tests/test_data"):
        for path in sorted(glob.glob(os.path.join(rh.REFERENCE_ROOT, sub, "*.yaml"))):
            if isinstance(rh.load_design(path), dict) and "platform" in rh.load_design(path):
                out.append(os.path.relpath(path, rh.REFERENCE_ROOT))
    return out


@pytest.mark.parametrize("deck", _platform_decks())
def test_every_solvable_deck_of_the_reference_tree(deck, oracle_ctx):
    import io
    from raft_amd import dropin
    d = rh.prepare_design(rh.load_design(os.path.join(rh.REFERENCE_ROOT, deck)), settings=dict(min_freq=0.01, max_freq=0.25))
    d["platform"].pop("outFolderQTF", None)

    def build():
        m = rh.build_model(d)
        for f in m.fowtList:
            f.outFolderQTF = None
            if f.nDOF != 6:
                cm = np.zeros((f.nDOF, f.nDOF))
                cm[:6, :6] = rh.DEFAULT_C_MOOR
                f.C_moor = cm
        return m

    case = rh.make_case(Hs=5.0, Tp=11.0, heading=25.0)
    # decks name their coefficient files relative to where upstream runs them from: the deck's own directory (its tests),
    # the tree's root (its examples)
    here = os.getcwd()
    err = None
    for cwd in (os.path.dirname(os.path.join(rh.REFERENCE_ROOT, deck)), rh.REFERENCE_ROOT):
        try:
            os.chdir(cwd)
            with contextlib.redirect_stdout(io.StringIO()):
                m_old = build()
                m_new = copy.deepcopy(m_old)
                Xi_old = m_old.solveDynamics(copy.deepcopy(case)).copy()
            err = None
            break
        except Exception as e:                                # noqa: BLE001 -- decks the reference itself cannot run in this container
            err = e
        finally:
            os.chdir(here)
    if err is not None:
        pytest.skip("reference cannot build/solve %s here: %s: %s" % (deck, type(err).__name__, str(err)[:80]))
    eng = dropin.Engine(oracle_ctx, qtf_backend=_numpy_qtf_backend)
    with contextlib.redirect_stdout(io.StringIO()):
        Xi_new = eng.solveDynamics(m_new, copy.deepcopy(case)).copy()
    assert Xi_new.shape == Xi_old.shape
    general = m_old.fowtList[0].nDOF != 6
    assert rel_err(Xi_new, Xi_old) < (1e-8 if general else 1e-10), deck
    for fn, fo in zip(m_new.fowtList, m_old.fowtList):
        assert rel_err(fn.B_hydro_drag, fo.B_hydro_drag) < 1e-9
        assert rel_err(fn.Xi_fullDOF, fo.Xi_fullDOF) < (1e-8 if general else 1e-10)


# ---------------------------------------------------------------------------------------------------------------------
# submerged rotors (raft_fowt.py:1861-1883).  The reference tree's one deck with an underwater turbine
# (designs/RM1_Floating.yaml) cannot be built here, so a live VolturnUS-S unit gets its rotor moved below the surface:
# hub position, rotated 6 x 6 inertial-excitation matrix with a force block and a (traceless) moment block.
def _sink_rotor(m):
    from raft.helpers import getH
    rng = np.random.default_rng(3)
    for f in m.fowtList:
        rot = f.rotorList[0]
        rot.r3 = np.array([f.x_ref + 4.0, f.y_ref - 3.0, -18.0])
        th = 0.3
        rot.R_q = np.array([[np.cos(th), -np.sin(th), 0], [np.sin(th), np.cos(th), 0], [0, 0, 1.0]]) @ \
            np.array([[1, 0, 0], [0, np.cos(0.2), -np.sin(0.2)], [0, np.sin(0.2), np.cos(0.2)]])
        I3 = np.zeros((3, 3))
        off = np.zeros((3, 3))
        for _ in range(3):                                   # three "blades": symmetric inertia translated from their offsets
            A = rng.normal(size=(3, 3))
            Ii = 2e5 * (A @ A.T)
            r = rng.uniform(-20, 20, size=3)
            I3 += Ii
            off += Ii @ getH(r)                              # translateMatrix3to6DOF's off-diagonal block (helpers.py:537-560)
        I6 = np.zeros((6, 6))
        I6[:3, :3], I6[:3, 3:], I6[3:, :3] = I3, off, off.T
        rot.I_hydro = I6


def _inject_bem(m, seed=9):
    """potential-flow coefficients on live units, the way FOWT.readHydro leaves them (raft_fowt.py:1479-1501): A_BEM, B_BEM
    in the first six reduced DOFs, X_BEM [nHeadings, nDOF, nw] in the first six rows, a heading grid -- smooth in w"""
    rng = np.random.default_rng(seed)
    for f in m.fowtList:
        n, nw = int(f.nDOF), f.nw
        w = np.asarray(f.w)
        sym = lambda a: 0.5 * (a + a.T)
        f.potModMaster = 2
        f.BEM_headings = np.array([0.0, 90.0, 180.0, 270.0])
        f.A_BEM = np.zeros([n, n, nw])
        f.B_BEM = np.zeros([n, n, nw])
        A0, B0 = sym(rng.uniform(0, 1, (6, 6))) + 2 * np.eye(6), sym(rng.uniform(0, 1, (6, 6))) + np.eye(6)
        scale = np.outer([1e6, 1e6, 1e6, 1e8, 1e8, 1e8], [1, 1, 1, 1e2, 1e2, 1e2]) ** 0.5 * 3.0
        f.A_BEM[:6, :6, :] = (A0 * scale)[:, :, None] / (1.0 + (w / 0.8) ** 2)[None, None, :]
        f.B_BEM[:6, :6, :] = (B0 * scale)[:, :, None] * ((w / 0.6) / (1.0 + (w / 0.6) ** 2))[None, None, :]
        f.X_BEM = np.zeros([4, n, nw], dtype=complex)
        amp = np.array([2e6, 2e6, 1e6, 4e7, 4e7, 1e7])[None, :, None] * rng.uniform(0.5, 1.5, (4, 6, 1))
        f.X_BEM[:, :6, :] = amp * np.exp(1j * (rng.uniform(0, 6, (4, 6, 1)) + 2.0 * w[None, None, :])) / (1.0 + (w / 0.7) ** 2)


@pytest.mark.parametrize("deck", ["tests/test_data/VolturnUS-S-flexible.yaml", "examples/VolturnUS-S_example.yaml"])
def test_potential_flow_coefficients_on_flexible_and_rigid_units(patch, deck):
    """A_BEM, B_BEM, X_BEM on a unit with MORE than 6 reduced DOFs (upstream lumps them at the first six DOFs,
    raft_fowt.py:1479-1501, 1796-1849; the loop raft_model.py:1019-1089 is nDOF-agnostic) -- and on the rigid unit for
    comparison: calcHydroExcitation's F_BEM(_fullDOF) and the whole solveDynamics against the NumPy path."""
    settings = dict(min_freq=0.01, max_freq=0.3, nIter=8, XiStart=0.1)
    case = rh.make_case(Hs=[5.0, 2.5], Tp=[11.0, 8.0], heading=[25.0, -40.0], spectrum=["JONSWAP"] * 2, gamma=[0, 0])
    m_new, m_old = _model(deck, settings), _model(deck, settings)
    for m in (m_new, m_old):
        for f in m.fowtList:
            f.potSecOrder = 0
            if f.nDOF != 6:
                cm = np.zeros((f.nDOF, f.nDOF))
                cm[:6, :6] = rh.DEFAULT_C_MOOR
                f.C_moor = cm
        _inject_bem(m)
    fn, fo = m_new.fowtList[0], m_old.fowtList[0]
    fn.calcHydroExcitation(copy.deepcopy(case), memberList=fn.memberList)
    with patch.unpatched():
        fo.calcHydroExcitation(copy.deepcopy(case), memberList=fo.memberList)
    assert np.any(fo.F_BEM) and rel_err(fn.F_BEM, fo.F_BEM) < 1e-10 and rel_err(fn.F_BEM_fullDOF, fo.F_BEM_fullDOF) < 1e-10
    assert rel_err(fn.F_hydro_iner, fo.F_hydro_iner) < 1e-10
    Xi_new = m_new.solveDynamics(copy.deepcopy(case)).copy()
    with patch.unpatched():
        Xi_old = m_old.solveDynamics(copy.deepcopy(case)).copy()
    general = fo.nDOF != 6
    assert Xi_new.shape == Xi_old.shape and rel_err(Xi_new, Xi_old) < (1e-8 if general else 1e-10)
    assert rel_err(fn.B_hydro_drag, fo.B_hydro_drag) < 1e-8 and rel_err(fn.Z, fo.Z) < 1e-10
    # ... and the coefficients matter in this set-up
    m_ref = _model(deck, settings)
    for f in m_ref.fowtList:
        f.potSecOrder = 0
        if f.nDOF != 6:
            cm = np.zeros((f.nDOF, f.nDOF))
            cm[:6, :6] = rh.DEFAULT_C_MOOR
            f.C_moor = cm
    with patch.unpatched():
        Xi_none = m_ref.solveDynamics(copy.deepcopy(case)).copy()
    assert rel_err(Xi_none, Xi_old) > 1e-2


def test_array_units_with_different_bem_heading_grids(patch):
    """VERDICT r5 missing 5: the units of an array each interpolate between the neighbours of their OWN BEM heading grid
    (raft_fowt.py:1804-1831); the patched path groups the units by grid -- one raftx_bem_excitation launch per distinct grid --
    and must give the NumPy path's F_BEM and coupled response, grids of different sizes and a wave heading in the wrap-around
    segment of one of them included."""
    settings = dict(min_freq=0.01, max_freq=0.3, nIter=8, XiStart=0.1)
    case = rh.make_case(Hs=[5.0, 2.5], Tp=[11.0, 8.0], heading=[25.0, -40.0], spectrum=["JONSWAP"] * 2, gamma=[0, 0])
    deck = "designs/VolturnUS-S_farm.yaml"
    m_new, m_old = _model(deck, settings), _model(deck, settings)
    grids = [np.array([0.0, 90.0, 180.0, 270.0]), np.array([10.0, 70.0, 130.0, 190.0, 250.0, 310.0])]
    for m in (m_new, m_old):
        assert len(m.fowtList) >= 2
        for f in m.fowtList:
            f.potSecOrder = 0
        _inject_bem(m)
        rng = np.random.default_rng(21)
        for i, f in enumerate(m.fowtList):
            g_ = grids[i % 2]
            if len(g_) != len(f.BEM_headings):                           # a six-heading grid: two more slabs, same construction
                X6 = np.zeros([len(g_), f.X_BEM.shape[1], f.nw], dtype=complex)
                X6[:4] = f.X_BEM
                X6[4:] = f.X_BEM[:2] * rng.uniform(0.6, 1.4, (2, 1, 1)) * np.exp(1j * rng.uniform(0, 6, (2, 1, 1)))
                f.X_BEM = X6
            f.BEM_headings = g_.copy()
    for fn, fo in zip(m_new.fowtList, m_old.fowtList):
        fn.calcHydroExcitation(copy.deepcopy(case), memberList=fn.memberList)
        with patch.unpatched():
            fo.calcHydroExcitation(copy.deepcopy(case), memberList=fo.memberList)
        assert np.any(fo.F_BEM) and rel_err(fn.F_BEM, fo.F_BEM) < 1e-10
    Xi_new = m_new.solveDynamics(copy.deepcopy(case)).copy()
    with patch.unpatched():
        Xi_old = m_old.solveDynamics(copy.deepcopy(case)).copy()
    assert Xi_new.shape == Xi_old.shape and rel_err(Xi_new, Xi_old) < 1e-9
    for fn, fo in zip(m_new.fowtList, m_old.fowtList):
        assert rel_err(fn.F_BEM, fo.F_BEM) < 1e-10 and rel_err(fn.Z, fo.Z) < 1e-10


def _second_sunk_rotor(m):
    """a second submerged rotor on every unit: a copy of the first one somewhere else, with another inertia matrix"""
    for f in m.fowtList:
        rot = copy.copy(f.rotorList[0])
        rot.r3 = np.array([f.x_ref - 6.0, f.y_ref + 5.0, -11.0])
        rot.I_hydro = 0.7 * np.asarray(f.rotorList[0].I_hydro).copy()
        rot.I_hydro[:3, :3] += np.diag([3e5, 1e5, 2e5])
        f.rotorList.append(rot)


@pytest.mark.parametrize("deck,two", [("designs/VolturnUS-S.yaml", True), ("designs/VolturnUS-S_farm.yaml", False),
                                      ("designs/VolturnUS-S_farm.yaml", True)])
def test_several_submerged_rotors_and_arrays(patch, deck, two):
    """raft_fowt.py:1861-1883 for more than one submerged rotor on a unit (each rotor's PRP-referred vector into its own
    node's slots of the full-DOF array, their sum in the reduced vector) and for the units of an array (one rotor table per
    unit beside the unit tables): FOWT.calcHydroExcitation and Model.solveDynamics against the NumPy path."""
    settings = dict(min_freq=0.01, max_freq=0.3, nIter=6, XiStart=0.1)
    case = rh.make_case(Hs=[4.0, 2.0], Tp=[10.0, 8.0], heading=[10.0, 55.0], spectrum=["JONSWAP"] * 2, gamma=[0, 0])
    m_new, m_old = _model(deck, settings), _model(deck, settings)
    for m in (m_new, m_old):
        _sink_rotor(m)
        if two:
            _second_sunk_rotor(m)
        if len(m.fowtList) > 1:                               # an array: the last unit keeps its rotor in the air
            for rot in m.fowtList[-1].rotorList:
                rot.r3 = np.array([rot.r3[0], rot.r3[1], 150.0])
    for fn, fo in zip(m_new.fowtList, m_old.fowtList):
        fn.calcHydroExcitation(copy.deepcopy(case), memberList=fn.memberList)
        with patch.unpatched():
            fo.calcHydroExcitation(copy.deepcopy(case), memberList=fo.memberList)
        assert rel_err(fn.F_hydro_iner, fo.F_hydro_iner) < 1e-10
        assert rel_err(fn.F_hydro_iner_fullDOF, fo.F_hydro_iner_fullDOF) < 1e-10
    Xi_new = m_new.solveDynamics(copy.deepcopy(case)).copy()
    with patch.unpatched():
        Xi_old = m_old.solveDynamics(copy.deepcopy(case)).copy()
    assert Xi_new.shape == Xi_old.shape and group_rel_err(Xi_new[:2], Xi_old[:2]) < 1e-10
    for fn, fo in zip(m_new.fowtList, m_old.fowtList):
        assert rel_err(fn.F_hydro_iner, fo.F_hydro_iner) < 1e-10 and rel_err(fn.B_hydro_drag, fo.B_hydro_drag) < 1e-9


@pytest.mark.parametrize("headings", [[15.0], [0.0, 40.0]])
def test_submerged_rotor_excitation_and_solve_match_the_reference(patch, headings):
    """FOWT.calcHydroExcitation and Model.solveDynamics with a submerged rotor: the device path evaluates the rotor's
    inertial excitation with pseudo-strips (one for the force block, couples for the moment block) and reproduces
    upstream's behaviour of adding it to the LAST heading only (:1868-1883), the full-DOF slots of the rotor node and the
    rotor's own kinematics arrays included."""
    settings = dict(min_freq=0.01, max_freq=0.3, nIter=8, XiStart=0.1)
    case = rh.make_case(Hs=5.0, Tp=11.0, heading=headings[0])
    if len(headings) > 1:
        case.update(wave_heading=headings, wave_spectrum=["JONSWAP"] * 2, wave_period=[11.0, 9.0], wave_height=[5.0, 2.0],
                    wave_gamma=[0, 0])
    m_new, m_old = _model("designs/VolturnUS-S.yaml", settings), _model("designs/VolturnUS-S.yaml", settings)
    _sink_rotor(m_new)
    _sink_rotor(m_old)
    fn, fo = m_new.fowtList[0], m_old.fowtList[0]
    fn.calcHydroExcitation(copy.deepcopy(case), memberList=fn.memberList)
    with patch.unpatched():
        fo.calcHydroExcitation(copy.deepcopy(case), memberList=fo.memberList)
        base = copy.deepcopy(fo)
        base.rotorList[0].r3 = np.array([0.0, 0.0, 150.0])   # the same unit with its rotor in the air
        base.calcHydroExcitation(copy.deepcopy(case), memberList=base.memberList)
    assert rel_err(fo.F_hydro_iner[-1], base.F_hydro_iner[-1]) > 1e-3          # the rotor term is not small in this set-up
    assert rel_err(fn.F_hydro_iner, fo.F_hydro_iner) < 1e-10
    assert rel_err(fn.F_hydro_iner_fullDOF, fo.F_hydro_iner_fullDOF) < 1e-10
    rn, ro = fn.rotorList[0], fo.rotorList[0]
    assert rel_err(rn.u, ro.u) < 1e-12 and rel_err(rn.ud, ro.ud) < 1e-12 and rel_err(rn.pDyn, ro.pDyn) < 1e-12
    Xi_new = m_new.solveDynamics(copy.deepcopy(case)).copy()
    with patch.unpatched():
        Xi_old = m_old.solveDynamics(copy.deepcopy(case)).copy()
    assert Xi_new.shape == Xi_old.shape
    assert group_rel_err(Xi_new[:len(headings)], Xi_old[:len(headings)]) < 1e-10
    assert rel_err(fn.F_hydro_iner, fo.F_hydro_iner) < 1e-10 and rel_err(fn.B_hydro_drag, fo.B_hydro_drag) < 1e-9
