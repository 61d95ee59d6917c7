"""CPU suite: the C-ABI boundary (header <-> shared objects) and the multi-rank sweep drivers, world_size 2, over
both host transports: torch.distributed gloo (tests/torch_comm.py) and the product's own TCP hub
(raft_amd.comm.HostComm, the channel that also carries the RCCL unique id on the GPUs).  No GPU compute: the HIP
library is only loaded and its symbols checked; the sharded runs use the CPU oracle as the per-rank backend."""
import ctypes
import os
import re
import socket

import numpy as np
import pytest

from raft_amd import sweep as sw
from raft_amd._abi import EXPORTS
from tests.conftest import HIP_SO, ORACLE_SO, ROOT
from raft_amd import snapshot as standin


def _header_functions():
    src = open(os.path.join(ROOT, "include", "raftx.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(raftx_[a-z_0-9]+)\s*\(", src)))


def test_header_and_binding_agree():
    names = _header_functions()
    assert len(names) >= 15
    assert set(names) == set(EXPORTS), set(names) ^ set(EXPORTS)


@pytest.mark.parametrize("path", [HIP_SO, ORACLE_SO])
def test_shared_objects_export_every_header_symbol(path, oracle_lib):
    """libraftx_hip.so must load on a GPU-less host and export the whole header (no compute call)."""
    assert os.path.exists(path), "%s missing: run __graft_entry__.build()" % path
    lib = ctypes.CDLL(path)
    for name in _header_functions():
        assert hasattr(lib, name), "%s does not export %s" % (path, name)
    lib.raftx_version.restype = ctypes.c_int
    lib.raftx_is_device.restype = ctypes.c_int
    assert lib.raftx_version() == 100
    assert lib.raftx_is_device() == (1 if path == HIP_SO else 0)


def test_product_does_not_import_torch():
    """PyTorch is plumbing of bench.py's launcher contract and of these tests only: nothing under raft_amd/ imports it
    (the multi-GPU exchange steps are the library's own RCCL binding, raft_amd/comm.py)."""
    pkg = os.path.join(ROOT, "raft_amd")
    for fn in sorted(os.listdir(pkg)):
        if fn.endswith(".py"):
            src = open(os.path.join(pkg, fn)).read()
            assert not re.search(r"^\s*(import torch|from torch)", src, flags=re.M), fn


def test_product_fails_loudly_without_a_gpu():
    """No CPU fallback: on a host without an MI355X the device library refuses to create a context."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is visible")
    from raft_amd import backend
    from raft_amd._abi import RaftxError
    with pytest.raises(RaftxError):
        backend.hip_library().context(0)
    assert backend.hip_library().device_count() == 0          # what bench.py's rank launcher checks --gpus against


def test_shard_bounds_cover_and_balance():
    for n in (0, 1, 7, 8, 10000):
        for world in (1, 2, 3, 8):
            b = [sw.shard_bounds(n, r, world) for r in range(world)]
            assert b[0][0] == 0 and b[-1][1] == n
            assert all(b[i][1] == b[i + 1][0] for i in range(world - 1))
            sizes = [hi - lo for lo, hi in b]
            assert max(sizes) - min(sizes) <= 1


def _c3_sweep(n):
    fx = standin.load_fixture("c3_variants.npz")
    off = fx["strip_offsets"]
    return sw.Sweep(off[:n + 1], fx["strips"][:off[n]], fx["M0"][:n], fx["B0"][:n], fx["C0"][:n], fx["w"], fx["k"],
                    fx["depth"], fx["zeta"][None], fx["beta"][None], int(fx["nIter"]), float(fx["XiStart"])), fx


def test_take_is_a_pure_slice(oracle_lib):
    s, _ = _c3_sweep(5)
    ctx = oracle_lib.context(0)
    full = s.run(ctx)
    part = s.take(2, 5).run(ctx)
    ctx.close()
    assert np.array_equal(full["Xi"][2:5].view(np.uint64), part["Xi"].view(np.uint64))
    assert np.array_equal(full["niter"][2:5], part["niter"])


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


TRANSPORTS = ["gloo", "host"]


def _make_comm(kind, rank, world, port):
    """world_size-2 communicator of the requested kind for a spawned rank process"""
    if kind == "gloo":
        import torch.distributed as dist
        from tests.torch_comm import GlooComm
        os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        return GlooComm(dist)
    from raft_amd.comm import from_env
    comm, how = from_env(None, prefer="host", environ={"RANK": str(rank), "WORLD_SIZE": str(world), "MASTER_ADDR": "127.0.0.1",
                                                       "RAFTX_COMM_PORT": str(port)})
    assert how == "host-tcp"
    return comm


def _rank_main(rank, world, port, n, out_path, kind):
    from raft_amd._abi import RaftxLib
    comm = _make_comm(kind, rank, world, port)
    try:
        s, _ = _c3_sweep(n)
        cases = {"w": s.w, "k": s.k, "zeta": s.zeta, "beta": s.beta, "depth": s.depth} if rank == 0 else None
        cases = sw.broadcast_cases(cases, comm)                 # shared tables come from rank 0
        if rank != 0:                                            # prove the broadcast carried them
            s.w, s.k, s.zeta, s.beta = cases["w"], cases["k"], cases["zeta"], cases["beta"]
        ctx = RaftxLib(ORACLE_SO).context(0)
        res = sw.run_sharded(s, ctx, comm)
        st = sw.run_stats_sharded(s, ctx, comm)
        ctx.close()
        if rank == 0:
            np.savez(out_path, Xi=res["Xi"], niter=res["niter"], flags=res["flags"], std=st["std"])
    finally:
        comm.close()


@pytest.mark.parametrize("kind", TRANSPORTS)
def test_two_rank_sweep_matches_single_process(tmp_path, oracle_lib, kind):
    """world_size 2: broadcast of the case tables, design sharding, gather of responses and of statistics to rank 0 --
    bitwise identical to the single-process run, and equal to the live-reference vectors."""
    import torch.multiprocessing as mp
    n, world = 7, 2                      # odd count: ragged shards (4 + 3)
    out = str(tmp_path / "gathered.npz")
    mp.spawn(_rank_main, args=(world, _free_port(), n, out, kind), nprocs=world, join=True)
    got = np.load(out)
    s, fx = _c3_sweep(n)
    ctx = oracle_lib.context(0)
    ref = s.run(ctx)
    ref_std = s.run_stats(ctx)["std"]
    ctx.close()
    assert np.array_equal(got["std"].view(np.uint64), ref_std.view(np.uint64))
    assert got["Xi"].shape == ref["Xi"].shape == (n, 1, 1, 6, s.nw)
    assert np.array_equal(got["Xi"].view(np.uint64), ref["Xi"].view(np.uint64))
    assert np.array_equal(got["niter"], ref["niter"]) and np.array_equal(got["flags"], ref["flags"])
    from tests.util import group_rel_err
    for j, sol in enumerate(fx["solved"]):
        if j < n:
            assert group_rel_err(got["Xi"][j, 0, :1], sol["Xi"][:1]) < 1e-10
            assert int(got["niter"][j, 0]) == int(sol["units"][0]["niter"])


def _geometry_sweep(n):
    """The first n C3 variants as MEMBER DESCRIPTIONS (generated on the device / by the oracle at upload)."""
    import json
    from tests.util import volturnus_sweep
    fx = standin.load_fixture("c3_variants.npz")
    fg = standin.load_fixture("geom_units.npz")
    u0 = [u for u in fg["units"] if u["name"] == "C3-variant-0"][0]
    M_rna = np.asarray(u0["M_struc"]) - np.asarray(u0["M_struc_bare"])
    C_rest = np.asarray(u0["C_struc"]) - np.asarray(u0["C_struc_bare"]) + np.diag([7e4, 7e4, 0, 0, 0, 1e8])
    D = volturnus_sweep(json.loads(fg["c3_base_json"]), np.asarray(fx["scales"])[:n]).tables()
    return sw.GeometrySweep(D, np.repeat(M_rna[None], n, 0), np.asarray(fx["B0"])[:n], np.repeat(C_rest[None], n, 0),
                            fx["w"], fx["k"], float(fx["depth"]), fx["zeta"], fx["beta"], int(fx["nIter"]),
                            float(fx["XiStart"])), fx


def _geom_rank_main(rank, world, port, n, out_path, kind):
    from raft_amd._abi import RaftxLib
    comm = _make_comm(kind, rank, world, port)
    try:
        s, _ = _geometry_sweep(n)
        ctx = RaftxLib(ORACLE_SO).context(0)
        res = sw.run_sharded(s, ctx, comm)
        ctx.close()
        if rank == 0:
            np.savez(out_path, Xi=res["Xi"], niter=res["niter"])
    finally:
        comm.close()


@pytest.mark.parametrize("kind", TRANSPORTS)
def test_two_rank_geometry_sweep(tmp_path, oracle_lib, kind):
    """Designs given as member descriptions shard by design like packed ones: every rank generates its own block
    (no collective on the data path), rank 0 gathers; equal to the packed-table sweep and to the live reference."""
    import torch.multiprocessing as mp
    n, world = 5, 2
    out = str(tmp_path / "geom_gathered.npz")
    mp.spawn(_geom_rank_main, args=(world, _free_port(), n, out, kind), nprocs=world, join=True)
    got = np.load(out)
    s, fx = _geometry_sweep(n)
    assert s.take(1, 4).tables.n_design == 3 and s.take(1, 4).tables.station_off[0] == 0
    ctx = oracle_lib.context(0)
    one = s.run(ctx)
    ctx.close()
    assert np.array_equal(got["Xi"].view(np.uint64), one["Xi"].view(np.uint64))
    packed, _ = _c3_sweep(n)
    ctx = oracle_lib.context(0)
    ref = packed.run(ctx)
    ctx.close()
    from tests.util import group_rel_err
    assert np.array_equal(got["niter"], ref["niter"])
    assert group_rel_err(got["Xi"].reshape(-1, 6, s.nw), ref["Xi"].reshape(-1, 6, s.nw)) < 1e-10
    for j, sol in enumerate(fx["solved"]):
        if j < n:
            assert group_rel_err(got["Xi"][j, 0, :1], sol["Xi"][:1]) < 1e-10


def _variant_sweep(n):
    """The first n C3 variants as PARAMETERS of an edit program (descriptors written by the library)."""
    import json
    from raft_amd import geometry as G
    g, fx = _geometry_sweep(n)
    fg = standin.load_fixture("geom_units.npz")
    prog = G.volturnus_program(json.loads(fg["c3_base_json"]))
    return sw.VariantSweep(prog, G.volturnus_params(np.asarray(fx["scales"])[:n]), g.M0, g.B0, g.C0, fx["w"], fx["k"],
                           float(fx["depth"]), fx["zeta"], fx["beta"], int(fx["nIter"]), float(fx["XiStart"])), g


def _variant_rank_main(rank, world, port, n, out_path, kind):
    from raft_amd._abi import RaftxLib
    comm = _make_comm(kind, rank, world, port)
    try:
        s, _ = _variant_sweep(n)
        ctx = RaftxLib(ORACLE_SO).context(0)
        res = sw.run_sharded(s, ctx, comm)
        st = sw.run_stats_sharded(s, ctx, comm)
        ctx.close()
        if rank == 0:
            np.savez(out_path, Xi=res["Xi"], niter=res["niter"], std=st["std"])
    finally:
        comm.close()


@pytest.mark.parametrize("kind", TRANSPORTS)
def test_two_rank_variant_sweep(tmp_path, oracle_lib, kind):
    """Designs given as PARAMETERS of an edit program (VariantSweep) shard by design like the others: every rank expands
    and generates its own block (params[lo:hi], no collective on the data path), rank 0 gathers; bit-identical to the
    sweep fed with the host-expanded member descriptions."""
    import torch.multiprocessing as mp
    n, world = 5, 2
    out = str(tmp_path / "variant_gathered.npz")
    mp.spawn(_variant_rank_main, args=(world, _free_port(), n, out, kind), nprocs=world, join=True)
    got = np.load(out)
    s, g = _variant_sweep(n)
    assert s.take(1, 4).n_design == 3 and np.array_equal(s.take(1, 4).params, s.params[1:4])
    ctx = oracle_lib.context(0)
    one = g.run(ctx)
    std = g.run_stats(ctx)["std"]
    ctx.close()
    assert np.array_equal(got["Xi"].view(np.uint64), one["Xi"].view(np.uint64)) and np.array_equal(got["niter"], one["niter"])
    assert np.array_equal(got["std"].view(np.uint64), std.view(np.uint64))
    assert sw.shard_fingerprint(s, 0, 3) != sw.shard_fingerprint(s, 1, 4)            # the parameter rows are part of a shard's identity


def _qtf_sets():
    from raft_amd import qtf as rq
    fx = standin.load_fixture("refgold_qtf_VolturnUS-S.npz")
    f = standin.build_model(fx["model"]).fowtList[0]
    tab = rq.pack_qtf(f)
    w2, k2 = f.w1_2nd[:9], f.k1_2nd[:9]
    rng = np.random.default_rng(3)
    Xi = 0.1 * (rng.normal(size=(3, 6, 9)) + 1j * rng.normal(size=(3, 6, 9)))
    return [tab] * 3, Xi, np.array([0.0, 0.5, -1.0]), w2, k2, f


def _numpy_qtf(t, X, b, w, k, h, rho, g, Ms, kay, rows=None):
    """numpy oracle; rows=(off, stride) keeps only those rows and their Hermitian mirrors (zeros elsewhere), like
    raftx_qtf_slender_rows."""
    from oracle import qtf_oracle
    q = np.array([qtf_oracle.qtf_slender_body(t[i], X[i], b[i], w, k, h, rho, g, Ms[i]) for i in range(len(t))]
                 ).reshape(len(t), len(w), len(w), 6)
    if rows is not None:
        n = len(w)
        i1, i2 = np.meshgrid(np.arange(n), np.arange(n), indexing="ij")
        keep = (np.minimum(i1, i2) % rows[1]) == rows[0]              # entry (i1,i2) belongs to row min(i1,i2)
        q = q * keep[None, :, :, None]
    return q


def _qtf_rank_main(rank, world, port, out_path, kind):
    comm = _make_comm(kind, rank, world, port)
    try:
        tabs, Xi, beta, w2, k2, f = _qtf_sets()
        q = sw.run_qtf_sharded(_numpy_qtf, tabs, Xi, beta, w2, k2, f.depth, f.rho_water, f.g,
                               np.array([f.M_struc] * len(tabs)), comm=comm)
        if rank == 0:
            np.save(out_path, q)
    finally:
        comm.close()


def _qtf_rows_rank_main(rank, world, port, out_path, kind):
    comm = _make_comm(kind, rank, world, port)
    try:
        tabs, Xi, beta, w2, k2, f = _qtf_sets()
        q = sw.run_qtf_rows_sharded(_numpy_qtf, tabs[:1], Xi[:1], beta[:1], w2, k2, f.depth, f.rho_water, f.g,
                                    np.array([f.M_struc]), comm=comm)
        if rank == 0:
            np.save(out_path, q)
    finally:
        comm.close()


@pytest.mark.parametrize("kind", TRANSPORTS)
def test_two_rank_qtf_rows_of_one_matrix(tmp_path, kind):
    """ONE QTF split by interleaved rows over two ranks and summed onto rank 0."""
    import torch.multiprocessing as mp
    out = str(tmp_path / "qrows.npy")
    mp.spawn(_qtf_rows_rank_main, args=(2, _free_port(), out, kind), nprocs=2, join=True)
    got = np.load(out)
    tabs, Xi, beta, w2, k2, f = _qtf_sets()
    ref = _numpy_qtf(tabs[:1], Xi[:1], beta[:1], w2, k2, f.depth, f.rho_water, f.g, np.array([f.M_struc]), None)
    assert got.shape == ref.shape and np.array_equal(got.view(np.float64), ref.view(np.float64))


def _farm_rank_main(rank, world, port, out_path, kind):
    from raft_amd import dropin
    from raft_amd._abi import RaftxLib
    from tests.util import load_model_fixture, case_from_fixture
    comm = _make_comm(kind, rank, world, port)
    try:
        fx, model = load_model_fixture("c4_farm.npz")
        cases = [case_from_fixture(c) for c in fx["cases"][:2]]
        sweep = dropin.sweep_from_units(model, cases + cases[:1])            # 3 sea states: ragged 2 + 1
        ctx = RaftxLib(ORACLE_SO).context(0)
        res = sw.run_farm_sharded(sweep, ctx, 4, Cc=fx["coupling_C"][None], comm=comm)
        ctx.close()
        if rank == 0:
            np.savez(out_path, Xi=res["Xi"], niter=res["niter"])
    finally:
        comm.close()


@pytest.mark.parametrize("kind", TRANSPORTS)
def test_two_rank_farm_cases(tmp_path, kind):
    """ONE 4-unit farm, its sea states block-partitioned over two ranks and gathered along the case axis
    (SURVEY.md 8e, C4): equal to the live reference's coupled responses."""
    import torch.multiprocessing as mp
    from tests.util import load_model_fixture, group_rel_err
    out = str(tmp_path / "farm.npz")
    mp.spawn(_farm_rank_main, args=(2, _free_port(), out, kind), nprocs=2, join=True)
    got = np.load(out)
    fx, _ = load_model_fixture("c4_farm.npz")
    assert got["Xi"].shape[:2] == (1, 3) and got["niter"].shape == (4, 3)
    from tests.util import ref_headings
    for i, c in enumerate(list(fx["cases"])[:2] + list(fx["cases"])[:1]):
        Xr, nH = ref_headings(c)
        assert group_rel_err(got["Xi"][0, i, :nH], Xr) < 1e-10
        assert [int(got["niter"][u, i]) for u in range(4)] == [int(c["units"][u]["niter"]) for u in range(4)]


def _flex_units(n_unit):
    """n_unit variants of the reference's flexible deck (its own T, node tables and matrices; inertia scaled per unit so that
    the units differ) x two sea states"""
    from raft_amd import dropin
    from raft_amd.flex import FlexSweep
    from tests.util import load_model_fixture, case_from_fixture
    fx, model = load_model_fixture("flex_volturnus.npz")
    base = case_from_fixture(fx["cases"][0])
    s = dropin.flex_sweep_from_models([model], [base, dict(base, wave_height=3.0, wave_period=8.0)])
    u0 = s.units[0]
    from raft_amd.flex import FlexUnit
    units = [FlexUnit(u0.tables, u0.Tn, u0.M * (1.0 + 0.1 * i), u0.B, u0.C) for i in range(n_unit)]
    return FlexSweep(units, s.w, s.k, s.depth, s.zeta, s.beta, s.nIter, s.XiStart, s.tol)


def _flex_rank_main(rank, world, port, n_unit, out_path, kind):
    from raft_amd._abi import RaftxLib
    comm = _make_comm(kind, rank, world, port)
    try:
        ctx = RaftxLib(ORACLE_SO).context(0)
        res = sw.run_flex_sharded(_flex_units(n_unit), ctx, comm)
        ctx.close()
        if rank == 0:
            np.savez(out_path, Xi=res["Xi"], niter=res["niter"], flags=res["flags"], B_drag=res["B_drag"])
    finally:
        comm.close()


@pytest.mark.parametrize("kind,n_unit", [("gloo", 3), ("host", 3), ("host", 1)])
def test_two_rank_flexible_units(tmp_path, oracle_lib, kind, n_unit):
    """Units with flexible members block-partitioned over two ranks (2 + 1; 1 + 0: a rank without units still takes part in
    the gather), every rank running the whole fixed point of its units: identical to the single-process batch."""
    import torch.multiprocessing as mp
    out = str(tmp_path / "flex.npz")
    mp.spawn(_flex_rank_main, args=(2, _free_port(), n_unit, out, kind), nprocs=2, join=True)
    got = np.load(out)
    ctx = oracle_lib.context(0)
    ref = _flex_units(n_unit).run(ctx)
    ctx.close()
    assert got["Xi"].shape == ref["Xi"].shape == (n_unit, 2, 1, 150, ref["Xi"].shape[4])
    for key in ("Xi", "B_drag"):
        assert np.array_equal(got[key].view(np.float64), ref[key].view(np.float64)), key
    assert np.array_equal(got["niter"], ref["niter"]) and np.array_equal(got["flags"], ref["flags"])
    assert len(set(float(np.abs(got["Xi"][i]).sum()) for i in range(n_unit))) == n_unit          # the units really differ


@pytest.mark.parametrize("kind", TRANSPORTS)
def test_two_rank_qtf_sets(tmp_path, kind):
    """QTF sets sharded over two ranks (2 + 1) and gathered: identical to the single-process batch."""
    import torch.multiprocessing as mp
    out = str(tmp_path / "q.npy")
    mp.spawn(_qtf_rank_main, args=(2, _free_port(), out, kind), nprocs=2, join=True)
    got = np.load(out)
    tabs, Xi, beta, w2, k2, f = _qtf_sets()
    ref = _numpy_qtf(tabs, Xi, beta, w2, k2, f.depth, f.rho_water, f.g, np.array([f.M_struc] * 3), None)
    assert got.shape == ref.shape and np.array_equal(got.view(np.float64), ref.view(np.float64))


# ------------------------------------------------------------------ shard checkpoints: an interrupted sweep resumes
def _ckpt_rank_main(rank, world, port, n, ckpt, out_path, kind):
    from raft_amd._abi import RaftxLib
    comm = _make_comm(kind, rank, world, port)
    try:
        s, _ = _c3_sweep(n)
        ctx = RaftxLib(ORACLE_SO).context(0)
        st = sw.run_stats_sharded(s, ctx, comm, checkpoint_dir=ckpt, shards_per_rank=2)
        ctx.close()
        if rank == 0:
            np.savez(out_path, std=st["std"], niter=st["niter"], flags=st["flags"])
    finally:
        comm.close()


def test_sharded_statistics_resume_from_shard_files(tmp_path, oracle_lib, monkeypatch):
    """run_stats_sharded(checkpoint_dir=...): shards are written as they finish; a second run loads what is there and
    solves only what is missing (one shard file deleted, one replaced by a file of another partition); two ranks then
    complete the same directory.  Every variant returns the bits of the plain run."""
    import torch.multiprocessing as mp
    n = 9
    s, _ = _c3_sweep(n)
    ctx = oracle_lib.context(0)
    ref = s.run_stats(ctx)
    ckpt = str(tmp_path / "ckpt")
    solved = []
    orig = sw.Sweep.run_stats
    monkeypatch.setattr(sw.Sweep, "run_stats", lambda self, c, want_psd=False: (solved.append(self.n_design), orig(self, c, want_psd))[1])
    first = sw.run_stats_sharded(s, ctx, None, checkpoint_dir=ckpt, shards_per_rank=4)
    assert sorted(solved) == [2, 2, 2, 3] and len(os.listdir(ckpt)) == 4
    files = sorted(os.listdir(ckpt))
    os.remove(os.path.join(ckpt, files[1]))
    np.savez(os.path.join(ckpt, files[2]), lo=0, hi=1, std=np.zeros((1, 1, 6)), niter=np.zeros((1, 1), np.int32), flags=np.zeros((1, 1), np.int32))
    solved.clear()
    second = sw.run_stats_sharded(s, ctx, None, checkpoint_dir=ckpt, shards_per_rank=4)
    assert len(solved) == 2                                       # only the missing and the mismatching shard
    ctx.close()
    for got in (first, second):
        assert np.array_equal(got["std"].view(np.uint64), ref["std"].view(np.uint64))
        assert np.array_equal(got["niter"], ref["niter"]) and np.array_equal(got["flags"], ref["flags"])
    monkeypatch.undo()
    # two ranks, two shards each, over the same directory (same 4-shard partition: everything is already there)
    out = str(tmp_path / "two_rank.npz")
    mp.spawn(_ckpt_rank_main, args=(2, _free_port(), n, ckpt, out, "host"), nprocs=2, join=True)
    got = np.load(out)
    assert np.array_equal(got["std"].view(np.uint64), ref["std"].view(np.uint64)) and np.array_equal(got["niter"], ref["niter"])


def test_checkpoint_of_another_sweep_is_not_reused(tmp_path, oracle_lib, monkeypatch):
    """Same directory, same shard bounds, different sea state / solver settings: the stored shards carry a fingerprint of
    everything their statistics depend on and are recomputed, not silently returned."""
    n = 5
    s, _ = _c3_sweep(n)
    ctx = oracle_lib.context(0)
    ckpt = str(tmp_path / "ckpt")
    first = sw.run_stats_sharded(s, ctx, None, checkpoint_dir=ckpt, shards_per_rank=2)
    s2, _ = _c3_sweep(n)
    s2.zeta = s2.zeta * 1.5                                        # another sea state, same shapes
    ref2 = s2.run_stats(ctx)
    solved = []
    orig = sw.Sweep.run_stats
    monkeypatch.setattr(sw.Sweep, "run_stats", lambda self, c, want_psd=False: (solved.append(self.n_design), orig(self, c, want_psd))[1])
    second = sw.run_stats_sharded(s2, ctx, None, checkpoint_dir=ckpt, shards_per_rank=2)
    assert len(solved) == 2                                        # both shards again
    assert np.array_equal(second["std"].view(np.uint64), ref2["std"].view(np.uint64))
    assert not np.array_equal(second["std"], first["std"])
    solved.clear()
    s3, _ = _c3_sweep(n)
    s3.zeta = s3.zeta * 1.5
    s3.nIter += 1                                                  # a solver setting
    sw.run_stats_sharded(s3, ctx, None, checkpoint_dir=ckpt, shards_per_rank=2)
    assert len(solved) == 2
    ctx.close()


def test_host_transport_admits_only_the_ranks_of_its_job(tmp_path):
    """HostComm: a peer with the wrong job token, an out-of-range rank or a rank already taken is not admitted; payloads
    are .npy / JSON, never pickles; all_max and the dictionary broadcast round-trip."""
    import threading
    from raft_amd import comm as rc
    port = _free_port()
    res = {}

    def root():
        c = rc.HostComm(0, 2, "127.0.0.1", port, timeout=20.0, token=b"job-A")
        res["max"] = c.all_max(1.5)
        res["d"] = c.broadcast_arrays({"a": np.arange(4.0), "n": 3, "name": "x", "f": 0.25, "none": None})
        res["rows"] = c.gather_rows(np.zeros((1, 2)))
        c.close()

    t = threading.Thread(target=root)
    t.start()
    import time
    time.sleep(0.2)
    with pytest.raises((ConnectionError, OSError)):                # wrong token
        rc.HostComm(1, 2, "127.0.0.1", port, timeout=2.0, token=b"job-B")
    with pytest.raises((ConnectionError, OSError, ValueError)):   # out-of-range rank
        rc.HostComm(5, 2, "127.0.0.1", port, timeout=2.0, token=b"job-A")
    c1 = rc.HostComm(1, 2, "127.0.0.1", port, timeout=20.0, token=b"job-A")
    assert c1.all_max(7.25) == 7.25
    d = c1.broadcast_arrays(None)
    c1.gather_rows(np.ones((2, 2)))
    c1.close()
    t.join(30)
    assert res["max"] == 7.25 and res["rows"].shape == (3, 2)
    assert np.array_equal(d["a"], np.arange(4.0)) and d["n"] == 3 and d["name"] == "x" and d["f"] == 0.25 and d["none"] is None
    with pytest.raises(TypeError):
        rc._pack_dict({"bad": object()})
    import inspect
    assert "pickle" not in inspect.getsource(rc).replace("allow_pickle", "").replace("unpickled", "").replace("pickles", "")


def test_crossing_refuses_what_it_cannot_carry(oracle_ctx):
    """GeometrySweep.run_crossing / submit_crossing with frequency-dependent matrices or BEM excitation raise instead of
    returning statistics without those terms."""
    from raft_amd.sweep import GeometrySweep

    class T:                                                      # the check runs before the tables are touched
        n_design = 2
    nw = 4
    z = np.zeros((2, 6, 6))
    s = GeometrySweep(T(), z, z, z, np.linspace(0.1, 1, nw), np.linspace(0.01, 0.1, nw), 200.0, np.ones((1, 1, nw)), np.zeros((1, 1)),
                      4, 0.1, MBw=np.zeros((2, 2, 6, 6, nw)))
    with pytest.raises(ValueError, match="MBw"):
        s.run_crossing(oracle_ctx)
    with pytest.raises(ValueError, match="MBw"):
        s.submit_crossing(oracle_ctx, 0)
