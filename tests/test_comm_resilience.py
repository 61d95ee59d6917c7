"""The first multi-rank RCCL run must be survivable: communicator creation runs under a deadline, proves itself with a
probe reduction, and the ranks agree on the outcome before anyone uses it (raft_amd/comm.py RcclComm, from_env).
Here with stand-in contexts on two threads over the real rendezvous channel (no GPU)."""
import socket
import threading
import time
import types

import numpy as np
import pytest

from raft_amd import comm as rcomm


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


class _Bus:
    """what a working two-rank reduction needs: both contributions, then the sum on the root"""

    def __init__(self, world):
        self.vals, self.bar = {}, threading.Barrier(world)


class _Ctx:
    def __init__(self, rank, bus, init="ok", reduce="ok"):
        self.rank, self.bus, self.init, self.reduce = rank, bus, init, reduce
        self.rlib = types.SimpleNamespace(is_device=True)
        self.inited = False

    def comm_unique_id(self):
        return bytes(128)

    def comm_init(self, rank, world, uid):
        if self.init == "raise":
            raise RuntimeError("hipErrorInvalidDevice")
        if self.init == "hang":
            time.sleep(30)
        self.inited = True

    def comm_reduce_sum(self, buf, root):
        if self.reduce == "wrong":
            buf[...] = -1.0
            return
        self.bus.vals[self.rank] = buf.copy()
        self.bus.bar.wait(timeout=5)
        if self.rank == root:
            buf[...] = sum(self.bus.vals.values())

    def comm_destroy(self):
        pass


def _two_ranks(make_ctx, fallback, deadline="1.5"):
    port, out = _free_port(), {}

    def run(rank):
        env = {"RANK": str(rank), "WORLD_SIZE": "2", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port - 101)}
        try:
            out[rank] = rcomm.from_env(make_ctx(rank), prefer="rccl", environ=env, fallback=fallback)
        except Exception as e:                            # noqa: BLE001
            out[rank] = e
    import os
    os.environ["RAFTX_COMM_INIT_TIMEOUT"] = deadline
    try:
        ts = [threading.Thread(target=run, args=(r,)) for r in range(2)]
        [t.start() for t in ts]
        [t.join(20) for t in ts]
    finally:
        del os.environ["RAFTX_COMM_INIT_TIMEOUT"]
    assert not any(t.is_alive() for t in ts)
    return out


def _close(out):
    for v in out.values():
        if isinstance(v, tuple):
            v[0].close()


def test_communicator_that_works_is_used():
    bus = _Bus(2)
    out = _two_ranks(lambda r: _Ctx(r, bus), "error")
    try:
        assert all(isinstance(v, tuple) and v[1] == "rccl" for v in out.values()), out
    finally:
        _close(out)


def test_ranks_fall_back_together_when_init_returns_an_error():
    """ncclCommInitRank RETURNS an error on every rank (what RCCL does for several ranks on one GPU): with fallback="host"
    both ranks land on the host transport, and it works"""
    bus = _Bus(2)
    out = _two_ranks(lambda r: _Ctx(r, bus, init="raise"), "host")
    try:
        for v in out.values():
            assert isinstance(v, tuple) and v[0].kind == "host-tcp" and v[1].startswith("host-tcp (RCCL unavailable"), out
        res = {}
        ts = [threading.Thread(target=lambda r=r: res.__setitem__(r, out[r][0].reduce_sum(np.array([r + 1.0])))) for r in range(2)]
        [t.start() for t in ts]
        [t.join(10) for t in ts]
        assert res[0][0] == 3.0 and res[1] is None
    finally:
        _close(out)


@pytest.mark.parametrize("fault", ["raise", "hang"])
def test_a_call_that_never_returns_poisons_the_context_and_nobody_falls_back(fault):
    """rank 1 fails (or never returns) in ncclCommInitRank; rank 0 is then stuck inside the collective.  Both come out
    within the deadline and take the SAME branch -- and because a thread is still inside the library on the stuck
    rank's context, that branch is an error even with fallback="host": the context is marked poisoned and never reused."""
    bus = _Bus(2)
    ctxs = {}

    def make(r):
        ctxs[r] = _Ctx(r, bus, init=fault if r == 1 else "hang")
        return ctxs[r]
    out = _two_ranks(make, "host")
    assert all(isinstance(v, rcomm.CommTimeout.__mro__[1]) and "RCCL communicator" in str(v) for v in out.values()), out
    assert all(isinstance(v.__cause__, rcomm.CommTimeout) for v in out.values()), out
    assert getattr(ctxs[0], "poisoned", None) and "never returned" in ctxs[0].poisoned
    if fault == "raise":
        assert getattr(ctxs[1], "poisoned", None) is None        # rank 1's call returned (with an error): its context is fine


def test_ranks_fail_together_when_fallback_is_an_error():
    bus = _Bus(2)
    out = _two_ranks(lambda r: _Ctx(r, bus, init="raise" if r == 1 else "ok"), "error")
    assert all(isinstance(v, RuntimeError) and "RCCL communicator" in str(v) for v in out.values()), out


def test_probe_reduction_with_a_wrong_sum_is_not_trusted():
    bus = _Bus(2)
    out = _two_ranks(lambda r: _Ctx(r, bus, reduce="wrong"), "host")
    try:
        assert all(isinstance(v, tuple) and "probe" in v[1] for v in out.values()), out
    finally:
        _close(out)
