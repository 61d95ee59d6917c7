"""Batched sweeps of units with MORE than 6 reduced DOFs (flexible members): many units x many sea states per launch.

The reference solves such a unit one load case at a time (raft/raft_model.py:966-1302 with the nDOF x nDOF matrices of
raft/raft_fowt.py's T reduction; 150 DOFs for tests/test_data/VolturnUS-S-flexible.yaml), and so does the drop-in
(raft_amd/dropin.py Engine._solve_general).  Here the same fixed point runs for EVERY (unit, sea state) of a batch at
once:

  * every structural node with wet strips of every unit is one "design" of the strip kernels (arms about the node's own
    position, raft_member.py:1969-1976, 2046-2056) -- one launch per iteration for the whole batch;
  * node motions T_node Xi, the projections sum_u T_u^T B_u T_u (a GEMM per pair: MFMA tiles) and T^T F, the impedance solves
    of all units, cases and bins (one launch, grid = bins x systems) and the convergence test / relaxation of
    raft_model.py:1103,1133 per (unit, case) all run on the device (raftx_flex_solve, raft_amd/csrc/raftx_flex.h): the host
    only reduces the inertial excitation once, before the fixed point;
  * a pair that has converged keeps its response and drops out of the linearisation's effect (its rows are still swept --
    the launches cover the batch -- but its results are frozen), exactly as if it had been solved alone.

Units come from live ``FOWT`` objects (``FlexUnit.from_fowt``: the reference's own T, M_struc, C_struc, C_elast ... -- the
finite-element assembly stays upstream) or from arrays.  All units of a sweep share the frequency grid; they may differ
in everything else, including their number of nodes, but not in nDOF.
"""
import weakref

import numpy as np

from .strips import pack_fowt_nodes

from .hostblas import few_threads as _few_blas_threads          # at most eight BLAS threads for the small host products

WAVE_RHO, WAVE_G = 1025.0, 9.81        # hard-wired defaults of Member.calcHydroExcitation (raft_member.py:1940)


class FlexUnit:
    """What the fixed point needs of one unit: strip tables per wet structural node, the six rows of T of each of those
    nodes [nNode,6,nDOF], and the frequency-independent matrices M_lin, B_lin, C_lin [nDOF,nDOF] of raft_model.py:1045-1047."""

    def __init__(self, tables, Tn, M_lin, B_lin, C_lin):
        self.tables = list(tables)
        self.Tn = np.ascontiguousarray(Tn, dtype=np.float64)
        self.M, self.B, self.C = (np.ascontiguousarray(a, dtype=np.float64) for a in (M_lin, B_lin, C_lin))
        n = self.M.shape[0]
        if self.Tn.shape != (len(self.tables), 6, n) or self.M.shape != (n, n) or self.B.shape != (n, n) or self.C.shape != (n, n):
            raise ValueError("FlexUnit: Tn must be [nNode,6,nDOF] and M, B, C [nDOF,nDOF]")

    @property
    def n_dof(self):
        return self.M.shape[0]

    @classmethod
    def from_fowt(cls, fowt, memberList=None):
        """From a live unit after calcStatics / calcHydroConstants (and solveStatics for C_moor): the sums of
        raft_model.py:1045-1047.  Frequency-dependent turbine matrices are not carried (the batch has none)."""
        if getattr(fowt, "nrotors", 0) > 0 and (np.any(np.asarray(fowt.A_aero)) or np.any(np.asarray(fowt.B_aero))):
            raise ValueError("FlexUnit.from_fowt: frequency-dependent aerodynamic matrices are not carried by the batched flexible sweep")
        rows, tables = pack_fowt_nodes(fowt, fowt.memberList if memberList is None else memberList)
        T = np.asarray(fowt.T, dtype=float)
        Tn = np.array([T[r:r + 6, :] for r in rows]).reshape(len(rows), 6, T.shape[1])
        B_gyro = np.sum(fowt.B_gyro, axis=2) if np.ndim(fowt.B_gyro) == 3 else np.asarray(fowt.B_gyro)
        return cls(tables, Tn, fowt.M_struc + fowt.A_hydro_morison, fowt.B_struc + B_gyro,
                   fowt.C_struc + fowt.C_hydro + fowt.C_moor + fowt.C_elast)


class _Token:
    """Key of one sweep's page-locked arrays in a context's store (hashable, weak-referenceable, never compared by value)."""
    __slots__ = ("__weakref__",)


def _drop_buffers(ctx_ref, token_id):
    """weakref.finalize callback of a FlexSweep: give its page-locked arrays back when the sweep is garbage-collected
    without release(ctx) (the context may already be closed, or gone)."""
    ctx = ctx_ref()
    if ctx is None:
        return
    for a in ctx.__dict__.get("_flex_bufs", {}).pop(token_id, {}).values():
        try:
            ctx.free_pinned(a)
        except Exception:                                            # noqa: BLE001 -- closed context: its memory is already freed
            pass


class FlexSweep:
    """units: list of FlexUnit (equal nDOF); w, k [nw]; zeta [nCase,nHead,nw]; beta [nCase,nHead] (heading 0 drives the
    linearisation, raft_fowt.py:1910); settings nIter, XiStart, tol as Model.solveDynamics (raft_model.py:49-58,966)."""

    def __init__(self, units, w, k, depth, zeta, beta, nIter, XiStart, tol=0.01):
        self.units = list(units)
        if not self.units:
            raise ValueError("FlexSweep: no units")
        self.n = self.units[0].n_dof
        if any(u.n_dof != self.n for u in self.units):
            raise ValueError("FlexSweep: all units must have the same number of reduced DOFs")
        self.w = np.ascontiguousarray(w, dtype=np.float64)
        self.k = np.ascontiguousarray(k, dtype=np.float64)
        self.depth = float(depth)
        zeta, beta = np.asarray(zeta, dtype=np.float64), np.asarray(beta, dtype=np.float64)
        if zeta.ndim == 2:
            zeta, beta = zeta[None], beta[None]
        self.zeta, self.beta = np.ascontiguousarray(zeta), np.ascontiguousarray(beta)
        self.nIter, self.XiStart, self.tol = int(nIter), float(XiStart), float(tol)
        self._token = _Token()                                       # key of this sweep's page-locked arrays in a context's store

    def _pinned(self, ctx, name, shape, dtype=np.float64, fill=None):
        """A page-locked host array of this sweep on ``ctx``, kept across runs (raftx_host_alloc through ctx.pinned_empty; plain
        NumPy memory where the backend has no such thing).  Blocking calls with pageable arrays of a few MB are what the runtime
        pins and unpins behind the caller's back: on the GPU box every second or third run took 40-65 ms instead of 20 until
        the arrays were page-locked (scripts/prof_flex_batch.py)."""
        key = id(self._token)
        stores = self._store(ctx)
        if key not in stores:
            stores[key] = {}
            try:                                                     # a sweep dropped without release(ctx) frees its arrays too
                weakref.finalize(self._token, _drop_buffers, weakref.ref(ctx), key)
            except TypeError:                                        # a context type without weak references: release() / close() only
                pass
        store = stores[key]
        a = store.get(name)
        if a is None or a.shape != tuple(shape) or a.dtype != np.dtype(dtype):
            if a is not None and hasattr(ctx, "free_pinned"):
                try:
                    ctx.free_pinned(a)
                except ValueError:
                    pass
            try:
                a = ctx.pinned_empty(shape, dtype=dtype)
            except Exception:                                        # noqa: BLE001 -- e.g. a backend without page-locked memory
                a = np.empty(shape, dtype=dtype)
            store[name] = a
            if fill is not None:
                a[...] = fill
        return a

    @staticmethod
    def _store(ctx):
        # the arrays live WITH the context (they die with it: ctx.close() frees page-locked memory), keyed by the sweep's token
        try:
            return ctx.__dict__.setdefault("_flex_bufs", {})
        except AttributeError:
            return {}

    def release(self, ctx):
        """Frees the page-locked arrays this sweep holds on ``ctx`` (ctx.close() does it too)."""
        for a in self._store(ctx).pop(id(self._token), {}).values():
            try:
                ctx.free_pinned(a)
            except (ValueError, AttributeError):
                pass

    def run(self, ctx, want_Z=False, copy=True):
        """{"Xi": [nUnit,nCase,nHead,nDOF,nw], "niter", "flags" [nUnit,nCase] (1 = converged), "B_drag" [nUnit,nCase,nDOF,nDOF],
        "kernel_ms": (inertial excitation sweep, the fixed point's span on the device)}.  copy=False: Xi, B_drag (and Z) are
        views of this sweep's page-locked arrays, valid until its next run on ``ctx`` / release(ctx) / ctx.close()."""
        nD, nC, nH, nw, n = len(self.units), self.zeta.shape[0], self.zeta.shape[1], len(self.w), self.n
        tables = [t for u in self.units for t in u.tables]
        first = np.concatenate([[0], np.cumsum([len(u.tables) for u in self.units])]).astype(int)
        nN = len(tables)
        Z6 = np.zeros((nN, 6, 6))
        t_strip = 0.0
        ctx.upload_designs(tables, Z6, Z6, Z6, nw)
        ctx.upload_cases(self.w, self.k, self.depth, WAVE_RHO, WAVE_G, self.zeta, self.beta)
        # stacked node rows of T per unit: T2[d] [nNode_d * 6, nDOF]
        T2 = [u.Tn.reshape(-1, n) for u in self.units]
        # the units' own matrices: page-locked arrays kept across runs, REFILLED every run (a few MB: units edited between
        # runs, or an array reallocated after a shape change, must never leave stale rows behind)
        Tn = self._pinned(ctx, "Tn", (nN, 6, n))
        M = self._pinned(ctx, "M", (nD, n, n))
        B0 = self._pinned(ctx, "B", (nD, n, n))
        C0 = self._pinned(ctx, "C", (nD, n, n))
        np.concatenate([u.Tn for u in self.units], out=Tn)
        for d, u in enumerate(self.units):
            M[d], B0[d], C0[d] = u.M, u.B, u.C
        # inertial excitation of every heading, reduced: F_iner[d,c,h] = sum_u T_u^T F_u  (raft_fowt.py:1886-1888)
        Fn = ctx.excitation(out=self._pinned(ctx, "Fn", (nN, nC, nH, 6, nw), np.complex128))
        t_strip += ctx.last_kernel_ms()
        F_iner = self._pinned(ctx, "F_iner", (nD, nC, nH, n, nw), np.complex128)
        with _few_blas_threads():
            for d in range(nD):
                blk = Fn[first[d]:first[d + 1]]                      # [nNode, nC, nH, 6, nw]
                F_iner[d] = np.matmul(T2[d].T, blk.transpose(1, 2, 0, 3, 4).reshape(nC, nH, -1, nw))
        bufs = {"Xi": self._pinned(ctx, "Xi", (nD, nC, nH, n, nw), np.complex128), "B_drag": self._pinned(ctx, "B_drag", (nD, nC, n, n))}
        if want_Z:
            bufs["Z"] = self._pinned(ctx, "Z", (nD, nC, n, n, nw), np.complex128)
        out = ctx.flex_solve(first, Tn, M, B0, C0, F_iner, self.nIter, self.tol, self.XiStart, want_F=False, want_Z=want_Z, out=bufs)
        t_fix = ctx.last_kernel_ms()
        take = (lambda a: np.array(a)) if copy else (lambda a: a)
        res = {"Xi": take(out["Xi"]), "niter": out["niter"], "flags": out["flags"], "B_drag": take(out["B_drag"]), "kernel_ms": (t_strip, t_fix)}
        if want_Z:
            res["Z"] = take(out["Z"])
        return res
