"""GPU suite: BASELINE.json's configurations at their FULL sizes, checked through size-independent properties (the oracle
cannot run these sizes in seconds): batch-position independence, equality of repeated designs / sea states, agreement
of a sample with the live-reference goldens, conservation under permutation."""
import json

import numpy as np
import pytest

from raft_amd import dropin, geometry as G
from tests import standin
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
    """configs[3]: 4-unit farm (24-DOF block solve) x 50 sea states in two launches.  Sea states are a random
    permutation-with-repeats of the two the live reference solved: every copy must equal its original bit for bit and
    the originals the reference."""
    fx, model = load_model_fixture("c4_farm.npz")
    base_cases = [case_from_fixture(c) for c in fx["cases"]]
    rng = np.random.default_rng(50)
    pick = rng.integers(0, len(base_cases), size=50)
    pick[:2] = [0, 1]
    sweep = dropin.sweep_from_units(model, [base_cases[i] for i in pick])
    out = sweep.run_farm(hip_ctx, 4, Cc=fx["coupling_C"][None])
    assert out["Xi"].shape[:2] == (1, 50) and out["Xi"].shape[3] == 24
    first = {int(p): int(np.nonzero(pick == p)[0][0]) for p in set(pick.tolist())}
    for i, p in enumerate(pick):
        assert np.array_equal(out["Xi"][0, i].view(np.uint64), out["Xi"][0, first[int(p)]].view(np.uint64))
    for p, i in first.items():
        c = fx["cases"][p]
        nH = c["Xi"].shape[0] - 1
        assert group_rel_err(out["Xi"][0, i, :nH], c["Xi"][:nH]) < 1e-9


def test_c2_full_size_three_sea_states(hip_ctx):
    """configs[1]: VolturnUS-S_example, 3 sea states x 200 bins, one launch -- responses and statistics against the live
    reference (the deck's own size is the full size)."""
    fx, model = load_model_fixture("c2_volturnus.npz")
    single = [c for c in fx["cases"] if len(np.atleast_1d(c["case"]["wave_heading"])) == 1][:3]
    sweep = dropin.sweep_from_models([model], [case_from_fixture(c) for c in single])
    out = sweep.run(hip_ctx)
    assert out["Xi"].shape == (1, 3, 1, 6, 200)
    for i, c in enumerate(single):
        assert group_rel_err(out["Xi"][0, i, :1], c["Xi"][:1]) < 1e-9


def test_pipelined_boundary_is_bit_identical(hip_lib, hip_ctx):
    """sweep.Pipeline: designs cut into ragged blocks, three contexts (streams) working concurrently from Python
    threads, results written straight into one output array -- bit-identical to the single-launch sweep."""
    from raft_amd.sweep import Pipeline, GeometrySweep
    c3 = standin.load_fixture("c3_variants.npz")
    fg = standin.load_fixture("geom_units.npz")
    n = 333
    scales = np.random.default_rng(9).uniform(0.75, 1.25, size=(n, 5))
    u0 = [u for u in fg["units"] if u["name"] == "C3-variant-0"][0]
    M_rna = np.asarray(u0["M_struc"]) - np.asarray(u0["M_struc_bare"])
    C_rest = np.asarray(u0["C_struc"]) - np.asarray(u0["C_struc_bare"]) + np.diag([7e4, 7e4, 0, 0, 0, 1e8])
    tables = volturnus_sweep(json.loads(fg["c3_base_json"]), scales).tables()
    zeta = np.stack([np.asarray(c3["zeta"]), 0.5 * np.asarray(c3["zeta"])])
    beta = np.array([[0.0], [0.7]])
    sweep = GeometrySweep(tables, np.repeat(M_rna[None], n, 0), np.zeros((n, 6, 6)), np.repeat(C_rest[None], n, 0), c3["w"],
                          c3["k"], float(c3["depth"]), zeta, beta, int(c3["nIter"]), float(c3["XiStart"]))
    ref = sweep.run(hip_ctx)
    pipe = Pipeline(hip_lib, n_workers=3)
    try:
        got = pipe.run(sweep, n_chunks=7)
        pin = pipe.run(sweep, n_chunks=4, pinned=True)["Xi"].copy()       # page-locked landing buffer (raftx_host_alloc)
        st = pipe.run(sweep, n_chunks=5, fetch="stats")
    finally:
        pipe.close()
    assert np.array_equal(got["Xi"].view(np.uint64), ref["Xi"].view(np.uint64))
    assert np.array_equal(pin.view(np.uint64), ref["Xi"].view(np.uint64))
    assert np.array_equal(got["niter"], ref["niter"]) and np.array_equal(got["flags"], ref["flags"])
    want = np.sqrt(0.5 * np.sum(np.abs(ref["Xi"][:, :, 0]) ** 2, axis=3))
    want[:, :, 3:] *= 57.29577951308232
    assert rel_err(st["std"], want) < 1e-12
