"""GPU suite for the multi-GPU exchange steps (include/raftx.h raftx_comm_*, raft_amd/comm.py).  The GPU box has ONE
MI355X: the RCCL binding is exercised with a single-rank communicator (every entry point, real librccl calls), and the
whole sharded driver with two processes sharing the GPU -- RCCL refuses two ranks on one device, so that run must land on
the host transport AND say so."""
import os
import socket

import numpy as np
import pytest

from raft_amd import sweep as sw
from raft_amd import snapshot as standin

pytestmark = pytest.mark.gpu


def _c3_sweep(n):
    fx = standin.load_fixture("c3_variants.npz")
    off = fx["strip_offsets"]
    return sw.Sweep(off[:n + 1], fx["strips"][:off[n]], fx["M0"][:n], fx["B0"][:n], fx["C0"][:n], fx["w"], fx["k"],
                    fx["depth"], fx["zeta"][None], fx["beta"][None], int(fx["nIter"]), float(fx["XiStart"])), fx


def test_single_rank_rccl_communicator_every_entry_point(hip_ctx):
    uid = hip_ctx.comm_unique_id()
    assert len(uid) == 128 and any(uid)
    hip_ctx.comm_init(0, 1, uid)
    try:
        a = np.arange(12.0).reshape(3, 4)
        assert np.array_equal(hip_ctx.comm_broadcast(a.copy(), 0), a)
        rows = (np.arange(15.0).reshape(5, 3) + 1j * np.arange(15.0).reshape(5, 3)[::-1])
        got = hip_ctx.comm_gather_rows(rows, [5], 0)
        assert got.dtype == rows.dtype and np.array_equal(got, rows)
        ints = np.arange(6, dtype=np.int32).reshape(6, 1)
        assert np.array_equal(hip_ctx.comm_gather_rows(ints, [6], 0), ints)
        s, _ = _c3_sweep(5)
        s.solve(hip_ctx)
        Xi = hip_ctx.fetch_results(want_Xi=True)["Xi"]
        g = hip_ctx.comm_gather_xi([5], 0)
        assert g.shape == (5, 1, 6, s.nw) and np.array_equal(g.view(np.uint64), Xi.reshape(g.shape).view(np.uint64))
        pinned = hip_ctx.pinned_empty((5, 1, 6, s.nw))
        hip_ctx.comm_gather_xi([5], 0, out=pinned)
        assert np.array_equal(pinned.view(np.uint64), g.view(np.uint64))
        hip_ctx.free_pinned(pinned)
        b = np.linspace(0, 1, 1000)
        assert np.array_equal(hip_ctx.comm_reduce_sum(b.copy(), 0), b)
        from raft_amd._abi import RaftxError
        with pytest.raises(RaftxError):
            hip_ctx.comm_gather_xi([4], 0)                      # counts must match the resident batch
        with pytest.raises(RaftxError):
            hip_ctx.comm_init(0, 1, uid)                        # one communicator per ctx
    finally:
        hip_ctx.comm_destroy()
    hip_ctx.comm_init(0, 1, hip_ctx.comm_unique_id())           # a ctx can be re-bound after destroy
    hip_ctx.comm_destroy()


def test_exchange_steps_with_empty_shards_and_refused_arguments(hip_ctx):
    """What the first N-rank run can meet that the plain single-rank test does not: a rank with ZERO rows (more ranks than
    designs), an empty reduction, and arguments the local validation refuses -- every one through the path the N-rank call
    takes (local validation -> the ranks' status vote, comm_agree -> grouped send / receive), and the communicator must
    stay usable after each refusal (a refused step must not leave half a group behind)."""
    from raft_amd._abi import RaftxError, _ptr
    hip_ctx.comm_init(0, 1, hip_ctx.comm_unique_id())
    L = hip_ctx.rlib.lib
    try:
        empty = np.zeros((0, 7))
        g = hip_ctx.comm_gather_rows(empty, [0], 0)                       # a rank without rows: nothing sent, nothing received
        assert g.shape == (0, 7)
        assert hip_ctx.comm_reduce_sum(np.zeros(0), 0).size == 0          # empty reduction
        one = np.arange(3.0)
        counts = np.array([-1], dtype=np.int64)
        assert L.raftx_comm_gather_rows(hip_ctx._h, _ptr(one), _ptr(counts), 24, _ptr(one.copy()), 0) != 0      # negative count
        assert b"negative count" in L.raftx_last_error(hip_ctx._h)
        assert L.raftx_comm_gather_rows(hip_ctx._h, _ptr(one), None, 24, _ptr(one.copy()), 0) != 0               # no counts at all
        counts = np.array([1], dtype=np.int64)
        assert L.raftx_comm_gather_rows(hip_ctx._h, None, _ptr(counts), 24, _ptr(one.copy()), 0) != 0             # rows promised, none given
        assert L.raftx_comm_gather_rows(hip_ctx._h, _ptr(one), _ptr(counts), 24, None, 0) != 0                    # root without a landing area
        assert L.raftx_comm_gather_rows(hip_ctx._h, _ptr(one), _ptr(counts), 24, _ptr(one.copy()), 3) != 0       # root outside the communicator
        with pytest.raises(RaftxError):
            hip_ctx.comm_broadcast(np.zeros(4), 2)
        # ... and after all of that the communicator still works
        rows = np.arange(10.0).reshape(5, 2)
        assert np.array_equal(hip_ctx.comm_gather_rows(rows, [5], 0), rows)
        b = np.linspace(0, 1, 33)
        assert np.array_equal(hip_ctx.comm_reduce_sum(b.copy(), 0), b)
        # the sharded drivers with a rank that holds nothing: shard_bounds gives empty blocks when ranks outnumber designs
        assert [sw.shard_bounds(2, r, 4) for r in range(4)] == [(0, 1), (1, 2), (2, 2), (2, 2)]
    finally:
        hip_ctx.comm_destroy()


def test_single_rank_rccl_comm_object_drives_the_sharded_sweep(hip_ctx, oracle_ctx):
    """raft_amd.comm.RcclComm (world 1) through every driver call of raft_amd.sweep."""
    from raft_amd.comm import HostComm, RcclComm
    comm = RcclComm(hip_ctx, HostComm(0, 1))
    try:
        assert comm.kind == "rccl" and comm.world == 1
        s, _ = _c3_sweep(6)
        cases = comm.broadcast_arrays({"w": s.w, "zeta": s.zeta, "depth": s.depth})
        assert np.array_equal(cases["w"], s.w) and cases["depth"] == s.depth
        s.solve(hip_ctx)
        Xi = comm.gather_xi(hip_ctx)
        ref = s.run(oracle_ctx)
        from tests.util import group_rel_err
        assert group_rel_err(Xi.reshape(6, 6, s.nw), ref["Xi"].reshape(6, 6, s.nw)) < 1e-9
        st = hip_ctx.motion_stats(float(s.w[1] - s.w[0]))[0]
        assert np.array_equal(comm.gather_rows(st.reshape(6, 6)), st.reshape(6, 6))
        q = np.arange(24.0).reshape(2, 3, 4) * (1 + 2j)
        assert np.array_equal(comm.reduce_sum(q), q)
    finally:
        comm.close()


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _rank_main(rank, world, port, n, out_path, own_gpu=False):
    from raft_amd import backend
    from raft_amd.comm import from_env
    # own_gpu: one device per rank, RCCL or nothing; otherwise both ranks on the one GPU of the box (a rehearsal: RCCL
    # refuses two ranks on one device, the host transport takes over and says so)
    ctx = backend.hip_library().context(rank if own_gpu else 0)
    comm, how = from_env(ctx, prefer="rccl", fallback="error" if own_gpu else "host",
                         environ={"RANK": str(rank), "WORLD_SIZE": str(world), "MASTER_ADDR": "127.0.0.1", "RAFTX_COMM_PORT": str(port)})
    try:
        s, _ = _c3_sweep(n)
        extra = {}
        if own_gpu:                                             # every exchange step of raft_amd.comm over real 2-rank RCCL
            cases = comm.broadcast_arrays({"w": s.w, "zeta": s.zeta, "depth": s.depth} if rank == 0 else None)
            assert np.array_equal(cases["w"], s.w) and cases["depth"] == s.depth
            rows = np.arange(12.0).reshape(4, 3) + 100.0 * rank
            g = comm.gather_rows(rows[:3 + rank])
            tot = comm.reduce_sum(np.full((5, 2), 1.0 + rank) * (1 + 1j))
            assert abs(comm.all_max(10.0 + rank) - 11.0) == 0.0
            if rank == 0:
                extra = {"rows": g, "tot": tot}
        res = sw.run_sharded(s, ctx, comm)
        st = sw.run_stats_sharded(s, ctx, comm)
        if rank == 0:
            np.savez(out_path, Xi=res["Xi"], niter=res["niter"], std=st["std"], how=np.array(how), **extra)
    finally:
        comm.close()
        ctx.close()


def test_two_processes_on_one_gpu_shard_and_gather(tmp_path, hip_ctx):
    """Two rank processes, one GPU: designs sharded 4 + 3, responses and statistics gathered on rank 0, bit-identical to
    the single-process run.  from_env asks for RCCL; with both ranks on one device it reports what it used."""
    import multiprocessing as mp
    n = 7
    out = str(tmp_path / "g.npz")
    port = _free_port()
    mpc = mp.get_context("spawn")
    procs = [mpc.Process(target=_rank_main, args=(r, 2, port, n, out)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(300)
        assert p.exitcode == 0
    got = np.load(out)
    how = str(got["how"])
    assert how == "rccl" or how.startswith("host-tcp (RCCL unavailable"), how
    print("two ranks on one GPU used:", how)
    s, _ = _c3_sweep(n)
    one = s.run(hip_ctx)
    std = s.run_stats(hip_ctx)["std"]
    assert np.array_equal(got["Xi"].view(np.uint64), one["Xi"].view(np.uint64))
    assert np.array_equal(got["niter"], one["niter"]) and np.array_equal(got["std"].view(np.uint64), std.view(np.uint64))


def test_two_rank_rccl_over_two_gpus(tmp_path, hip_lib, hip_ctx):
    """Where two GPUs are visible: two rank processes, one device each, the library's RCCL communicator for real --
    broadcast, gather_rows, reduce_sum, gather_xi -- and the sharded C3 sweep bit-identical to the single-rank run.
    Skipped on one-GPU boxes (the build pool's); no host fallback is allowed here."""
    from raft_amd._abi import RaftxError
    try:
        hip_lib.context(1).close()
    except RaftxError:
        pytest.skip("one GPU visible")
    import multiprocessing as mp
    n = 9
    out = str(tmp_path / "g2.npz")
    port = _free_port()
    mpc = mp.get_context("spawn")
    procs = [mpc.Process(target=_rank_main, args=(r, 2, port, n, out, True)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(300)
        assert p.exitcode == 0
    got = np.load(out)
    assert str(got["how"]) == "rccl"
    rows = np.arange(12.0).reshape(4, 3)
    assert np.array_equal(got["rows"], np.concatenate([rows[:3], rows + 100.0], axis=0))
    assert np.array_equal(got["tot"], np.full((5, 2), 3.0) * (1 + 1j))
    s, _ = _c3_sweep(n)
    one = s.run(hip_ctx)
    std = s.run_stats(hip_ctx)["std"]
    assert np.array_equal(got["Xi"].view(np.uint64), one["Xi"].view(np.uint64))
    assert np.array_equal(got["niter"], one["niter"]) and np.array_equal(got["std"].view(np.uint64), std.view(np.uint64))


@pytest.mark.gpu
def test_device_count_is_what_the_rank_launcher_sees(oracle_lib):
    """raftx_device_count: bench.py --gpus N refuses to start more ranks than this (the oracle has no devices)."""
    from raft_amd import backend
    assert backend.hip_library().device_count() >= 1
    assert oracle_lib.device_count() == 0
