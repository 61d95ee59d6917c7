"""GPU suite: BASELINE.json's configurations at their FULL sizes, checked through size-independent properties (the oracle
cannot run these sizes in seconds): batch-position independence, equality of repeated designs / sea states, agreement
of a sample with the live-reference goldens, conservation under permutation."""
import json

import numpy as np
import pytest

from raft_amd import dropin, geometry as G
from raft_amd import snapshot as standin
from tests.util import group_rel_err, rel_err, case_from_fixture, load_model_fixture, volturnus_sweep, rao_group_err, \
    psd_group_err

pytestmark = pytest.mark.gpu


def test_c3_full_size_10k_designs_generated_on_device(hip_ctx):
    """configs[2]: 10 000 design variants x 200 bins.  The 64 reference-built variants are repeated 156.25 times through
    the device generator; every replica must reproduce its original bit for bit wherever it lands in the batch (workgroup
    -> XCD mapping, neighbours, ragged strip counts), the originals must match the live reference, and the response
    statistics of the batch must equal those recomputed from the downloaded responses."""
    c3 = standin.load_fixture("c3_variants.npz")
    fg = standin.load_fixture("geom_units.npz")
    nD = 10000
    idx = np.arange(nD) % 64
    scales = np.asarray(c3["scales"])[idx]
    u0 = [u for u in fg["units"] if u["name"] == "C3-variant-0"][0]
    M_rna = np.asarray(u0["M_struc"]) - np.asarray(u0["M_struc_bare"])
    C_rest = np.asarray(u0["C_struc"]) - np.asarray(u0["C_struc_bare"]) + np.diag([7e4, 7e4, 0, 0, 0, 1e8])
    D = volturnus_sweep(json.loads(fg["c3_base_json"]), scales).tables()
    off = hip_ctx.build_designs(D.member_off, D.members, D.station_off, D.stations, np.repeat(M_rna[None], nD, 0),
                                np.repeat(np.asarray(c3["B0"])[:1], nD, 0), np.repeat(C_rest[None], nD, 0), 200,
                                cap_off=D.cap_off, caps=D.caps, add_mask=7)
    ref_off = np.asarray(c3["strip_offsets"])
    assert np.array_equal(np.diff(off), np.diff(ref_off)[idx])
    hip_ctx.upload_cases(c3["w"], c3["k"], float(c3["depth"]), 1025.0, 9.81, np.asarray(c3["zeta"])[None],
                         np.asarray(c3["beta"])[None])
    hip_ctx.solve_dynamics_device(int(c3["nIter"]), 0.01, float(c3["XiStart"]))
    r = hip_ctx.fetch_results(want_Xi=True)
    Xi = r["Xi"].reshape(nD, 6, 200)
    assert not np.any(r["flags"] & 2)
    base = Xi[:64]
    for k in range(1, nD // 64):
        assert np.array_equal(Xi[64 * k:64 * (k + 1)].view(np.uint64), base.view(np.uint64)), k
    assert np.array_equal(r["niter"].reshape(-1)[:nD - nD % 64].reshape(-1, 64), np.tile(r["niter"].reshape(-1)[:64], (nD // 64, 1)))
    dw = float(c3["w"][1] - c3["w"][0])
    for j, sol in enumerate(c3["solved"]):
        assert int(r["niter"][j, 0]) == int(sol["units"][0]["niter"])
        assert group_rel_err(r["Xi"][j, 0, :1], np.asarray(sol["Xi"])[:1]) < 1e-9
        # the north-star metric: RAOs and motion PSDs, gate 1e-6, expected ~1e-12
        assert rao_group_err(r["Xi"][j, 0, 0], np.asarray(sol["Xi"])[0], np.asarray(c3["zeta"])[0]) < 1e-9
        assert psd_group_err(r["Xi"][j, 0, :1], np.asarray(sol["Xi"])[:1], dw) < 1e-9
    std, _ = hip_ctx.motion_stats(dw)
    want = np.sqrt(0.5 * np.sum(np.abs(Xi) ** 2, axis=2))
    want[:, 3:] *= 57.29577951308232
    assert rel_err(std.reshape(nD, 6), want) < 1e-12


def test_c4_full_size_four_units_fifty_sea_states(hip_ctx):
    """configs[3] at its real shape: 4-unit farm (24-DOF block solve) x 200 bins x the 50 seeded sea states of
    default_rng(1), in two launches (per-unit fixed points, coupled 24 x 24 solves) -- EVERY sea state against the live
    reference's own solveDynamics of the farm (tests/golden/c4_farm.npz), and position-independence: the same sea states
    in reversed order give bit-identical responses."""
    from tests.util import ref_headings
    fx, model = load_model_fixture("c4_farm.npz")
    cases = [case_from_fixture(c) for c in fx["cases"]]
    assert len(cases) == 50 and model.nw == 200
    sweep = dropin.sweep_from_units(model, cases)
    out = sweep.run_farm(hip_ctx, 4, Cc=fx["coupling_C"][None])
    assert out["Xi"].shape == (1, 50, 1, 24, 200)
    assert not np.any(out["flags"] & 2)
    for i, c in enumerate(fx["cases"]):
        Xr, nH = ref_headings(c)
        assert group_rel_err(out["Xi"][0, i, :nH], Xr) < 1e-9
        assert [int(out["niter"][u, i]) for u in range(4)] == [int(c["units"][u]["niter"]) for u in range(4)]
        zeta = np.asarray(c["units"][0]["zeta"])[0]
        for u in range(4):                                 # SURVEY 8d metric per unit: RAOs
            assert rao_group_err(out["Xi"][0, i, 0, 6 * u:6 * u + 6], Xr[0, 6 * u:6 * u + 6], zeta) < 1e-9
    rev = dropin.sweep_from_units(model, cases[::-1]).run_farm(hip_ctx, 4, Cc=fx["coupling_C"][None])
    assert np.array_equal(rev["Xi"][0, ::-1].view(np.uint64), out["Xi"][0].view(np.uint64))
    assert np.array_equal(rev["niter"][:, ::-1], out["niter"])


def test_c5_full_size_oc4semi_200x200_against_the_live_reference(hip_ctx):
    """configs[4] at its real shape: examples/OC4semi-RAFT_QTF.yaml (MacCamy-Fuchs columns with heave plates, inclined
    braces), nw = 200, second-order grid 200 x 200 (20 100 upper-triangle pairs), sea state (6 m, 12 s) at 0 and 30 deg:
    first-order fixed point, Kim & Yue tables, slender-body QTF from the converged motions, second-order force,
    restarted fixed point -- all on the device through the drop-in Model.solveDynamics, against the LIVE reference
    run of oracle/make_golden.py:fixture_c5_full (7 minutes of reference time per case)."""
    from tests.util import load_model_fixture
    fx, model = load_model_fixture("c5_oc4semi_full.npz")
    f = model.fowtList[0]
    assert model.nw == 200 and len(f.w1_2nd) == 200
    eng = dropin.Engine(hip_ctx)
    iu = np.triu_indices(200)
    for c in fx["cases"]:
        Xi = eng.solveDynamics(model, case_from_fixture(c))
        u = c["units"][0]
        nH = len(np.atleast_1d(u["beta"]))
        assert int(model._raftx_niter[0]) == int(u["niter"])
        q = f.qtf[:, :, 0, :]
        assert rel_err(q[iu], u["qtf_triu"]) < 1e-9
        off = ~np.eye(200, dtype=bool)                                  # Hermitian completion (raft_fowt.py:2069-2070)
        assert np.array_equal(q[off], np.conj(np.transpose(q, (1, 0, 2)))[off])
        assert rel_err(f.Fhydro_2nd, u["Fhydro_2nd"]) < 1e-9
        assert rel_err(f.Fhydro_2nd_mean, u["Fhydro_2nd_mean"]) < 1e-9
        assert group_rel_err(Xi[:nH], np.asarray(c["Xi"])[:nH]) < 1e-9
        assert rao_group_err(Xi[0], np.asarray(c["Xi"])[0], np.asarray(u["zeta"])[0]) < 1e-9
