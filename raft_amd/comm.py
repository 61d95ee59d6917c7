"""The once-per-batch exchange steps of a multi-GPU sweep: shared sea-state tables out, responses /
statistics / QTF partials back.  One process per GPU; nothing here runs while kernels run.

Two transports behind one small interface (``rank``, ``world``, ``broadcast_arrays``, ``gather_rows``, ``gather_xi``,
``reduce_sum``, ``barrier``, ``close``):

  * ``RcclComm`` -- the product path: the library's own RCCL communicator (include/raftx.h raftx_comm_*: ncclBroadcast,
    grouped ncclSend/ncclRecv gather-to-root, ncclReduce, over xGMI), bound to the rank's raftx context.  Responses are
    gathered straight from the HBM buffers they were solved into.  No PyTorch anywhere.
  * ``HostComm`` -- a TCP hub on MASTER_ADDR (rank 0 listens, the others connect): the rendezvous that carries the
    128-byte RCCL unique id, and the whole transport of the CPU tests (two processes, the oracle library) and of
    single-GPU rehearsals.  Plain sockets + NumPy buffers.

``from_env(ctx)`` builds the right one from RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT (the variables
``python -m torch.distributed.run`` exports), listening on MASTER_PORT + 101 (RAFTX_COMM_PORT overrides) so that it
does not collide with the launcher's own store.
"""
import os
import pickle
import socket
import struct
import time

import numpy as np


def _send_msg(sock, payload):
    sock.sendall(struct.pack("<Q", len(payload)))
    sock.sendall(payload)


def _recv_exact(sock, n):
    buf = bytearray(n)
    view = memoryview(buf)
    got = 0
    while got < n:
        k = sock.recv_into(view[got:], n - got)
        if k == 0:
            raise ConnectionError("peer closed the connection")
        got += k
    return bytes(buf)


def _recv_msg(sock):
    (n,) = struct.unpack("<Q", _recv_exact(sock, 8))
    return _recv_exact(sock, n)


class HostComm:
    """Star topology over TCP: rank 0 holds one socket per peer.  Collectives are rooted at rank 0."""

    kind = "host-tcp"

    def __init__(self, rank, world, addr="127.0.0.1", port=29601, timeout=120.0):
        self.rank, self.world = int(rank), int(world)
        self.peers = {}
        self.sock = None
        if self.world == 1:
            return
        if self.rank == 0:
            srv = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
            srv.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
            srv.bind((addr, int(port)))
            srv.listen(self.world)
            srv.settimeout(timeout)
            try:
                while len(self.peers) < self.world - 1:
                    conn, _ = srv.accept()
                    conn.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                    conn.settimeout(timeout)
                    (r,) = struct.unpack("<i", _recv_exact(conn, 4))
                    self.peers[r] = conn
            finally:
                srv.close()
        else:
            t_end = time.time() + timeout
            while True:
                try:
                    s = socket.create_connection((addr, int(port)), timeout=timeout)
                    break
                except OSError:
                    if time.time() > t_end:
                        raise
                    time.sleep(0.05)
            s.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
            s.settimeout(timeout)
            s.sendall(struct.pack("<i", self.rank))
            self.sock = s

    # -------------------------------------------------------------- primitives (root = 0)
    def bcast_bytes(self, payload=None):
        if self.world == 1:
            return payload
        if self.rank == 0:
            for r in range(1, self.world):
                _send_msg(self.peers[r], payload)
            return payload
        return _recv_msg(self.sock)

    def gather_bytes(self, payload):
        """list of every rank's payload on rank 0, None elsewhere"""
        if self.world == 1:
            return [payload]
        if self.rank == 0:
            return [payload] + [_recv_msg(self.peers[r]) for r in range(1, self.world)]
        _send_msg(self.sock, payload)
        return None

    def barrier(self):
        self.gather_bytes(b"")
        self.bcast_bytes(b"")

    # -------------------------------------------------------------- the interface the sweep drivers use
    def broadcast_arrays(self, arrays=None):
        """dict of arrays / scalars from rank 0 to every rank"""
        return pickle.loads(self.bcast_bytes(pickle.dumps(arrays, protocol=4) if self.rank == 0 else None))

    def all_counts(self, n):
        c = self.gather_bytes(struct.pack("<q", int(n)))
        c = pickle.loads(self.bcast_bytes(pickle.dumps([struct.unpack("<q", x)[0] for x in c]) if self.rank == 0 else None))
        return np.asarray(c, dtype=np.int64)

    def gather_rows(self, local, counts=None):
        """Row blocks (first axis) of every rank concatenated in rank order on rank 0; None elsewhere."""
        local = np.ascontiguousarray(local)
        parts = self.gather_bytes(pickle.dumps(local, protocol=4))
        if parts is None:
            return None
        return np.concatenate([pickle.loads(p) for p in parts], axis=0)

    def gather_xi(self, ctx, counts=None, out=None):
        """The resident responses of every rank's last solve, [sum pairs, nHead, 6, nw] on rank 0."""
        r = ctx.fetch_results(want_Xi=True)["Xi"]
        full = self.gather_rows(r.reshape((-1,) + r.shape[2:]))
        if full is not None and out is not None:
            out[...] = full.reshape(out.shape)
            return out
        return full

    def reduce_sum(self, arr):
        """element-wise sum over ranks on rank 0 (rank order: deterministic); None elsewhere"""
        parts = self.gather_bytes(pickle.dumps(np.ascontiguousarray(arr), protocol=4))
        if parts is None:
            return None
        total = pickle.loads(parts[0]).copy()
        for p in parts[1:]:
            total += pickle.loads(p)
        return total

    def close(self):
        for s in list(self.peers.values()) + ([self.sock] if self.sock else []):
            try:
                s.close()
            except OSError:
                pass
        self.peers, self.sock = {}, None


class RcclComm:
    """The library's RCCL communicator on ``ctx`` (raftx_comm_*); ``boot`` is the HostComm that carried the unique id and
    keeps serving the tiny control messages (counts)."""

    kind = "rccl"

    def __init__(self, ctx, boot):
        self.ctx, self.boot = ctx, boot
        self.rank, self.world = boot.rank, boot.world
        uid = None
        if self.rank == 0:
            try:
                uid = ctx.comm_unique_id()
            except Exception as e:                      # noqa: BLE001 -- tell the waiting ranks instead of leaving them blocked
                uid = b"ERR " + str(e).encode()[:200]
        uid = boot.bcast_bytes(uid)
        if len(uid) != 128:
            raise RuntimeError("no RCCL unique id from rank 0: %s" % uid.decode(errors="replace"))
        ctx.comm_init(self.rank, self.world, uid)

    def barrier(self):
        self.boot.barrier()

    def all_counts(self, n):
        return self.boot.all_counts(n)

    def broadcast_arrays(self, arrays=None):
        """Structure (names, shapes, dtypes, scalars) over the rendezvous channel, array payloads over RCCL."""
        meta = None
        if self.rank == 0:
            meta = {k: (np.asarray(v).shape, np.asarray(v).dtype.str) if isinstance(v, np.ndarray) else ("scalar", v)
                    for k, v in arrays.items()}
        meta = pickle.loads(self.boot.bcast_bytes(pickle.dumps(meta) if self.rank == 0 else None))
        out = {}
        for k in meta:
            if meta[k][0] == "scalar":
                out[k] = meta[k][1]
                continue
            shape, dt = meta[k]
            a = np.ascontiguousarray(arrays[k]) if self.rank == 0 else np.empty(shape, dtype=np.dtype(dt))
            self.ctx.comm_broadcast(a, 0)
            out[k] = a
        return out

    def gather_rows(self, local, counts=None):
        local = np.ascontiguousarray(local)
        if counts is None:
            counts = self.all_counts(local.shape[0])
        return self.ctx.comm_gather_rows(local, counts, 0)

    def gather_xi(self, ctx=None, counts=None, out=None):
        ctx = ctx or self.ctx
        if counts is None:
            counts = self.all_counts(ctx.nDesign * ctx.nCase)
        return ctx.comm_gather_xi(counts, 0, out=out)

    def reduce_sum(self, arr):
        a = np.ascontiguousarray(arr)
        is_c = np.iscomplexobj(a)
        buf = a.view(np.float64).copy() if is_c else a.astype(np.float64, copy=True)
        self.ctx.comm_reduce_sum(buf, 0)
        if self.rank != 0:
            return None
        return buf.view(np.complex128).reshape(a.shape) if is_c else buf.reshape(a.shape)

    def close(self):
        try:
            self.ctx.comm_destroy()
        finally:
            self.boot.close()


def from_env(ctx=None, prefer="rccl", environ=None):
    """(comm, kind) for this process from the launcher's environment.  prefer="rccl" needs a device context; if the
    RCCL communicator cannot be created (e.g. a rehearsal with several ranks on ONE GPU, which RCCL refuses), the host
    transport is used and ``kind`` says so -- the caller reports it, nothing is silent."""
    env = os.environ if environ is None else environ
    rank, world = int(env.get("RANK", "0")), int(env.get("WORLD_SIZE", "1"))
    addr = env.get("MASTER_ADDR", "127.0.0.1")
    port = int(env.get("RAFTX_COMM_PORT", int(env.get("MASTER_PORT", "29500")) + 101))
    boot = HostComm(rank, world, addr, port)
    if prefer == "rccl" and ctx is not None and ctx.rlib.is_device and world > 1:
        try:
            c = RcclComm(ctx, boot)
            return c, c.kind
        except Exception as e:          # noqa: BLE001 -- ncclCommInitRank is collective: every rank lands here together
            return boot, "host-tcp (RCCL unavailable: %s)" % str(e)[:120]
    return boot, boot.kind
