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
import hashlib
import hmac
import io
import json
import os
import socket
import struct
import sys
import threading
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


# Wire format: arrays travel as .npy bytes (np.save / np.load with allow_pickle=False), dictionaries of arrays and
# scalars as a JSON header followed by the arrays -- nothing on this channel is ever unpickled.
def _pack_array(a):
    f = io.BytesIO()
    np.save(f, np.ascontiguousarray(a), allow_pickle=False)
    return f.getvalue()


def _unpack_array(b):
    return np.load(io.BytesIO(b), allow_pickle=False)


def _pack_dict(d):
    head, blobs = {}, []
    for k, v in d.items():
        if isinstance(v, np.ndarray):
            head[k] = ["array", len(blobs)]
            blobs.append(_pack_array(v))
        elif isinstance(v, (bool, int, float, str)) or v is None:
            head[k] = ["scalar", v]
        elif isinstance(v, (np.integer, np.floating)):
            head[k] = ["scalar", v.item()]
        else:
            raise TypeError("broadcast_arrays carries arrays and plain scalars, not %s (%r)" % (type(v).__name__, k))
    h = json.dumps(head).encode()
    return b"".join([struct.pack("<QQ", len(h), len(blobs)), h] + [struct.pack("<Q", len(x)) + x for x in blobs])


def _unpack_dict(b):
    nh, nb = struct.unpack_from("<QQ", b, 0)
    at = 16
    head = json.loads(b[at:at + nh].decode())
    at += nh
    blobs = []
    for _ in range(nb):
        (n,) = struct.unpack_from("<Q", b, at)
        blobs.append(b[at + 8:at + 8 + n])
        at += 8 + n
    return {k: (_unpack_array(blobs[v[1]]) if v[0] == "array" else v[1]) for k, v in head.items()}


def _is_loopback(addr):
    return addr in ("localhost", "::1") or addr.startswith("127.")


def _job_token(env, addr, port, world):
    """Key of the admission handshake.  RAFTX_COMM_TOKEN, if the launcher exports one (bench.py's own rank launcher
    draws 128 random bits per job), is a secret and the handshake then authenticates the peers.  Without it the key is
    derived from what every rank of the job sees -- the rendezvous endpoint, the world size and the elastic run id: all
    guessable, so the handshake is then only a guard against ACCIDENTAL cross-job connections (two jobs on one port),
    which is all a loopback rendezvous on a single node needs.  A rendezvous on a routable address must bring a
    secret: ``HostComm`` refuses to listen there without one."""
    tok = env.get("RAFTX_COMM_TOKEN")
    if tok:
        return tok.encode()
    if world > 1 and not _is_loopback(str(addr)):
        raise RuntimeError("rendezvous on %s:%s is reachable from other hosts: export RAFTX_COMM_TOKEN (a shared secret) "
                           "for the ranks of the job" % (addr, port))
    return ("raftx|%s|%s|%s|%s" % (addr, port, world, env.get("TORCHELASTIC_RUN_ID", ""))).encode()


class HostComm:
    """Star topology over TCP: rank 0 holds one socket per peer.  Collectives are rooted at rank 0.

    A peer answers a nonce with its HMAC under the job token (``_job_token``: an authentication when the launcher
    supplied a secret, a cross-job collision guard otherwise) and announces a rank in 1 .. world-1 that nobody else has
    taken before it is admitted; a connection gets 2 s for that exchange, so a stranger cannot hold up the rendezvous.  ``timeout`` bounds the rendezvous only; collectives
    block (ranks of a sweep may finish minutes apart -- a resumed rank only loads its shards) unless
    ``collective_timeout`` / RAFTX_COMM_TIMEOUT sets a bound."""

    kind = "host-tcp"

    def __init__(self, rank, world, addr="127.0.0.1", port=29601, timeout=120.0, token=None, collective_timeout=None):
        self.rank, self.world = int(rank), int(world)
        self.peers = {}
        self.sock = None
        if self.world == 1:
            return
        if not 0 <= self.rank < self.world:
            raise ValueError("rank %d outside 0..%d" % (self.rank, self.world - 1))
        token = token if token is not None else _job_token(os.environ, addr, port, self.world)
        if collective_timeout is None and os.environ.get("RAFTX_COMM_TIMEOUT"):
            collective_timeout = float(os.environ["RAFTX_COMM_TIMEOUT"])
        if self.rank == 0:
            srv = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
            srv.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
            srv.bind((addr, int(port)))
            srv.listen(self.world)
            t_end = time.time() + timeout
            try:
                while len(self.peers) < self.world - 1:
                    srv.settimeout(max(t_end - time.time(), 0.01))
                    conn, _ = srv.accept()
                    try:
                        conn.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                        conn.settimeout(2.0)
                        nonce = os.urandom(16)
                        conn.sendall(nonce)
                        (r,) = struct.unpack("<i", _recv_exact(conn, 4))
                        mac = _recv_exact(conn, 32)
                        good = hmac.compare_digest(mac, hmac.new(token, nonce + struct.pack("<i", r), hashlib.sha256).digest())
                        if not good or not 0 < r < self.world or r in self.peers:
                            conn.close()                  # a stranger, an out-of-range or a duplicate rank: not admitted
                            continue
                        conn.sendall(b"OK")
                        conn.settimeout(collective_timeout)
                        self.peers[r] = conn
                    except (OSError, ConnectionError, struct.error):
                        conn.close()
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
            nonce = _recv_exact(s, 16)
            s.sendall(struct.pack("<i", self.rank) + hmac.new(token, nonce + struct.pack("<i", self.rank), hashlib.sha256).digest())
            if _recv_exact(s, 2) != b"OK":
                raise ConnectionError("rank 0 did not admit rank %d" % self.rank)
            s.settimeout(collective_timeout)
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

    def all_max(self, x):
        """max of a float over the ranks, on every rank (the elapsed time of a timed region)"""
        parts = self.gather_bytes(struct.pack("<d", float(x)))
        m = struct.pack("<d", max(struct.unpack("<d", p)[0] for p in parts)) if parts is not None else None
        return struct.unpack("<d", self.bcast_bytes(m))[0]

    def gather_floats(self, xs):
        """[world, len(xs)] array of every rank's floats on rank 0 (per-rank timings of a bench), None elsewhere"""
        xs = [float(x) for x in np.atleast_1d(xs)]
        parts = self.gather_bytes(struct.pack("<%dd" % len(xs), *xs))
        if parts is None:
            return None
        return np.array([struct.unpack("<%dd" % len(xs), p) for p in parts])

    # -------------------------------------------------------------- the interface the sweep drivers use
    def broadcast_arrays(self, arrays=None):
        """dict of arrays / scalars from rank 0 to every rank"""
        return _unpack_dict(self.bcast_bytes(_pack_dict(arrays) if self.rank == 0 else None))

    def all_counts(self, n):
        c = self.gather_bytes(struct.pack("<q", int(n)))
        b = self.bcast_bytes(b"".join(c) if self.rank == 0 else None)
        return np.frombuffer(b, dtype="<i8").astype(np.int64)

    def gather_rows(self, local, counts=None):
        """Row blocks (first axis) of every rank concatenated in rank order on rank 0; None elsewhere."""
        parts = self.gather_bytes(_pack_array(local))
        if parts is None:
            return None
        return np.concatenate([_unpack_array(p) for p in parts], axis=0)

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
        parts = self.gather_bytes(_pack_array(arr))
        if parts is None:
            return None
        total = _unpack_array(parts[0]).copy()
        for p in parts[1:]:
            total += _unpack_array(p)
        return total

    def close(self):
        for s in list(self.peers.values()) + ([self.sock] if self.sock else []):
            try:
                s.close()
            except OSError:
                pass
        self.peers, self.sock = {}, None


def _attempt(fn, seconds):
    """None if fn() returned within the deadline, else what went wrong.  fn runs on a daemon thread (ctypes releases the
    GIL inside the library): a call that never returns is abandoned, not waited for."""
    box = {}

    def run():
        try:
            fn()
        except BaseException as e:                      # noqa: BLE001 -- reported to the caller
            box["e"] = "%s: %s" % (type(e).__name__, e)
    t = threading.Thread(target=run, daemon=True)
    t.start()
    t.join(seconds)
    if t.is_alive():
        return CommTimeout("no return within %.0f s" % seconds)
    return box.get("e")


class CommTimeout(RuntimeError):
    """A library call of the communicator's set-up did not return: its thread is still INSIDE the library on that context.
    The library is not thread-safe per context, so the context is poisoned (``ctx.poisoned``) -- nothing may use it again,
    neither for exchange steps nor for uploads / solves -- and there is no fallback to the host transport on it."""


class RcclComm:
    """The library's RCCL communicator on ``ctx`` (raftx_comm_*); ``boot`` is the HostComm that carried the unique id and
    keeps serving the tiny control messages (counts)."""

    kind = "rccl"

    def __init__(self, ctx, boot, deadline=None):
        self.ctx, self.boot = ctx, boot
        self.rank, self.world = boot.rank, boot.world
        if deadline is None:
            deadline = float(os.environ.get("RAFTX_COMM_INIT_TIMEOUT", "180"))
        uid = None
        if self.rank == 0:
            try:
                uid = ctx.comm_unique_id()
            except Exception as e:                      # noqa: BLE001 -- tell the waiting ranks instead of leaving them blocked
                uid = b"ERR " + str(e).encode()[:200]
        uid = boot.bcast_bytes(uid)
        if len(uid) != 128:
            raise RuntimeError("no RCCL unique id from rank 0: %s" % uid.decode(errors="replace"))
        # ncclCommInitRank is collective and blocks: a rank that fails (or never arrives) leaves the others inside it.  It
        # runs under a deadline, and the ranks AGREE on the outcome over the rendezvous channel before anyone uses the
        # communicator -- so that they fail, or fall back, together.
        self._agree(self._guard(_attempt(lambda: ctx.comm_init(self.rank, self.world, uid), deadline)), "ncclCommInitRank")
        # one probe collective before the communicator is trusted with results: 1 + 2 + .. + world on the root
        probe = np.array([self.rank + 1.0])
        err = self._guard(_attempt(lambda: ctx.comm_reduce_sum(probe, 0), deadline))
        if err is None and self.rank == 0 and probe[0] != self.world * (self.world + 1) / 2:
            err = "probe reduction gave %r, expected %r" % (probe[0], self.world * (self.world + 1) / 2)
        self._agree(err, "probe ncclReduce")

    def _guard(self, err):
        if isinstance(err, CommTimeout):                 # the abandoned thread may still be running in the library on this ctx
            self.ctx.poisoned = "a communicator set-up call never returned (%s)" % err
        return err

    def _agree(self, err, what):
        # 2 = some rank timed out (its context is unusable: nobody falls back), 1 = some rank got an error back, 0 = fine
        bad = self.boot.all_max(2.0 if isinstance(err, CommTimeout) else (1.0 if err else 0.0))
        if bad >= 2:
            raise CommTimeout("%s: %s" % (what, err or "another rank's call never returned"))
        if bad:
            raise RuntimeError("%s: %s" % (what, err or "another rank failed"))

    def barrier(self):
        self.boot.barrier()

    def all_max(self, x):
        return self.boot.all_max(x)

    def all_counts(self, n):
        return self.boot.all_counts(n)

    def gather_floats(self, xs):
        return self.boot.gather_floats(xs)

    def broadcast_arrays(self, arrays=None):
        """Structure (names, shapes, dtypes, scalars) over the rendezvous channel, array payloads over RCCL."""
        meta = None
        if self.rank == 0:
            meta = {k: [list(np.asarray(v).shape), np.asarray(v).dtype.str] if isinstance(v, np.ndarray)
                    else ["scalar", v.item() if isinstance(v, (np.integer, np.floating)) else v] for k, v in arrays.items()}
        meta = json.loads(self.boot.bcast_bytes(json.dumps(meta).encode() if self.rank == 0 else None).decode())
        out = {}
        for k in meta:
            if meta[k][0] == "scalar":
                out[k] = meta[k][1]
                continue
            shape, dt = tuple(meta[k][0]), meta[k][1]
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


def from_env(ctx=None, prefer="rccl", environ=None, fallback="error"):
    """(comm, kind) for this process from the launcher's environment.  prefer="rccl" needs a device context.  The RCCL
    communicator is created under a deadline (RAFTX_COMM_INIT_TIMEOUT, 180 s), proves itself with one probe reduction,
    and the ranks agree on the outcome over the rendezvous channel -- so they land in the same branch below together.
    fallback="error" (default): a sweep whose ranks own distinct GPUs must not quietly move its exchange steps onto TCP.
    fallback="host": the host transport is used and ``kind`` says so, with the reason (rehearsals with several ranks on
    ONE GPU, which RCCL refuses; bench.py, whose JSON line carries ``kind``) -- but only after an error that RETURNED:
    after a timeout (CommTimeout) a thread is still inside the library on ``ctx``, the context is marked ``poisoned`` and
    every rank raises."""
    env = os.environ if environ is None else environ
    rank, world = int(env.get("RANK", "0")), int(env.get("WORLD_SIZE", "1"))
    addr = env.get("MASTER_ADDR", "127.0.0.1")
    port = int(env.get("RAFTX_COMM_PORT", int(env.get("MASTER_PORT", "29500")) + 101))
    boot = HostComm(rank, world, addr, port, token=_job_token(env, addr, port, world))
    if prefer == "rccl" and ctx is not None and ctx.rlib.is_device and world > 1:
        try:
            c = RcclComm(ctx, boot)
            return c, c.kind
        except Exception as e:          # noqa: BLE001 -- every rank lands here together
            if fallback != "host" or isinstance(e, CommTimeout):     # a timed-out context is never reused, not even over TCP
                boot.close()
                raise RuntimeError("RCCL communicator of rank %d / %d could not be created: %s" % (rank, world, e)) from e
            sys.stderr.write("raftx comm: rank %d / %d falls back to the host transport: %s\n" % (rank, world, e))
            return boot, "host-tcp (RCCL unavailable: %s)" % str(e)[:160]
    return boot, boot.kind
