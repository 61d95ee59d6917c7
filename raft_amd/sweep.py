"""Batched (design x sea state) sweeps and their sharding over the GPUs of a node.

The reference runs the sweep as nested Python loops in one process
(raft/parametersweep.py:39-100; raft/omdao_raft.py:746-792 builds one Model per
optimiser iterate; raft/raft_model.py:290-337 loops the cases).  Here a sweep is
one launch of the fused kernel per rank:

  * ``Sweep`` holds the packed strip tables + 6x6 matrices of the designs and
    the sea states they are all solved for;
  * ``Sweep.shard(rank, world)`` block-partitions the DESIGN axis -- (design,
    case) items are independent, nothing is exchanged while solving;
  * ``run_sharded`` is the one-process-per-GPU driver: rank 0 broadcasts the
    shared case tables (w, k, zeta, beta: a few KB), every rank solves its own
    designs, and the responses are gathered to rank 0 through a raft_amd.comm
    communicator: the library's own RCCL communicator over xGMI on the GPUs
    (responses leave from the HBM buffers they were solved into), a TCP hub in
    the CPU tests.  No collective sits inside the solve.
"""
import numpy as np

from ._abi import NFIELD

WAVE_RHO, WAVE_G = 1025.0, 9.81        # hard-wired defaults of Member.calcHydroExcitation (raft_member.py:1940)


def shard_bounds(n, rank, world):
    """Contiguous block partition of range(n): the first n % world ranks get one extra item."""
    if world < 1 or not (0 <= rank < world):
        raise ValueError("bad rank/world %r/%r" % (rank, world))
    base, rem = divmod(int(n), int(world))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


class Sweep:
    """Packed inputs of a (design x case) batch.

    off [nD+1], strips [nS,32], M0/B0/C0 [nD,6,6], optional MBw [nD,2,6,6,nw],
    optional (cmoff [nD+1], cm [nRows,2,nw]); cases: w,k [nw], zeta [nC,nH,nw],
    beta [nC,nH]; settings nIter, tol, XiStart (raft_model.py:49-58,966)."""

    def __init__(self, off, strips, M0, B0, C0, w, k, depth, zeta, beta, nIter, XiStart,
                 tol=0.01, MBw=None, cmoff=None, cm=None, rho=1025.0, g=9.81):
        self.off = np.ascontiguousarray(off, dtype=np.int64)
        self.strips = np.ascontiguousarray(strips, dtype=np.float64).reshape(-1, NFIELD)
        self.M0, self.B0, self.C0 = (np.ascontiguousarray(a, dtype=np.float64) for a in (M0, B0, C0))
        self.w = np.ascontiguousarray(w, dtype=np.float64)
        self.k = np.ascontiguousarray(k, dtype=np.float64)
        self.depth, self.rho, self.g = float(depth), float(rho), float(g)
        zeta = np.asarray(zeta, dtype=np.float64)
        beta = np.asarray(beta, dtype=np.float64)
        if zeta.ndim == 2:
            zeta, beta = zeta[None], beta[None]
        self.zeta, self.beta = np.ascontiguousarray(zeta), np.ascontiguousarray(beta)
        self.nIter, self.XiStart, self.tol = int(nIter), float(XiStart), float(tol)
        self.MBw = None if MBw is None else np.ascontiguousarray(MBw, dtype=np.float64)
        self.cmoff = None if cm is None else np.ascontiguousarray(cmoff, dtype=np.int64)
        self.cm = None if cm is None else np.ascontiguousarray(cm, dtype=np.complex128)
        if len(self.off) - 1 != self.M0.shape[0]:
            raise ValueError("strip offsets describe %d designs, matrices %d" % (len(self.off) - 1, self.M0.shape[0]))

    # ------------------------------------------------------------------ sizes
    @property
    def n_design(self):
        return len(self.off) - 1

    @property
    def n_case(self):
        return self.zeta.shape[0]

    @property
    def n_head(self):
        return self.zeta.shape[1]

    @property
    def nw(self):
        return len(self.w)

    @classmethod
    def from_fowts(cls, fowts_mats, w, k, depth, zeta, beta, nIter, XiStart, tol=0.01):
        """fowts_mats: iterable of (StripTable, M0, B0, C0[, MBw]) -- e.g. from raft_amd.strips.pack_fowt
        and the sums of raft_model.py:1045-1047."""
        tables = [t[0] for t in fowts_mats]
        off = np.concatenate([[0], np.cumsum([t.n for t in tables])]).astype(np.int64)
        strips = np.concatenate([t.strips for t in tables], axis=0) if tables else np.zeros((0, NFIELD))
        M0 = np.array([t[1] for t in fowts_mats])
        B0 = np.array([t[2] for t in fowts_mats])
        C0 = np.array([t[3] for t in fowts_mats])
        MBw = None
        if any(len(t) > 4 and t[4] is not None for t in fowts_mats):
            nw = len(w)
            MBw = np.array([t[4] if len(t) > 4 and t[4] is not None else np.zeros((2, 6, 6, nw)) for t in fowts_mats])
        cmoff = cm = None
        if any(t.cm_mcf is not None for t in tables):
            cmoff = np.concatenate([[0], np.cumsum([0 if t.cm_mcf is None else t.cm_mcf.shape[0] for t in tables])])
            cm = np.concatenate([t.cm_mcf for t in tables if t.cm_mcf is not None], axis=0)
        return cls(off, strips, M0, B0, C0, w, k, depth, zeta, beta, nIter, XiStart, tol, MBw, cmoff, cm)

    # ------------------------------------------------------------------ potential-flow excitation
    def set_bem(self, headings_deg, X_BEM, heading_adjust=None, xy_ref=None):
        """Potential-flow excitation coefficients of the designs, X_BEM [nD,nHeadBEM,6,nw] in the wave-heading frame as
        FOWT.readHydro leaves them (raft_amd/bem.py read_hydro): ``upload`` then evaluates F_BEM for every (design,
        case, heading) on the device (raftx_bem_excitation, raft_fowt.py:1796-1849) and the solves use it as their
        F_extra -- nothing of size nCase x nHead x 6 x nw is built or uploaded by the host.  The frequency-dependent
        A_BEM / B_BEM enter through MBw."""
        self.bem = dict(heads=np.ascontiguousarray(headings_deg, dtype=np.float64),
                        X=np.ascontiguousarray(X_BEM, dtype=np.complex128),
                        hadj=None if heading_adjust is None else np.ascontiguousarray(heading_adjust, dtype=np.float64),
                        xy=None if xy_ref is None else np.ascontiguousarray(xy_ref, dtype=np.float64))
        return self

    def _take_bem(self, dst, lo, hi):
        b = getattr(self, "bem", None)
        if b is not None:
            dst.bem = dict(heads=b["heads"], X=b["X"][lo:hi], hadj=None if b["hadj"] is None else b["hadj"][lo:hi],
                           xy=None if b["xy"] is None else b["xy"][lo:hi])
        return dst

    def _upload_bem(self, ctx):
        b = getattr(self, "bem", None)
        if b is not None:
            ctx.bem_excitation(b["heads"], b["X"], b["hadj"], b["xy"])

    # ------------------------------------------------------------------ sharding
    def take(self, lo, hi):
        """The sub-sweep of designs [lo, hi) (all cases)."""
        s0, s1 = self.off[lo], self.off[hi]
        cmoff = cm = None
        if self.cm is not None:
            c0, c1 = self.cmoff[lo], self.cmoff[hi]
            cmoff, cm = self.cmoff[lo:hi + 1] - c0, self.cm[c0:c1]
        return self._take_bem(Sweep(self.off[lo:hi + 1] - s0, self.strips[s0:s1], self.M0[lo:hi], self.B0[lo:hi], self.C0[lo:hi],
                                    self.w, self.k, self.depth, self.zeta, self.beta, self.nIter, self.XiStart, self.tol,
                                    None if self.MBw is None else self.MBw[lo:hi], cmoff, cm, self.rho, self.g), lo, hi)

    def shard(self, rank, world):
        return self.take(*shard_bounds(self.n_design, rank, world))

    def take_cases(self, lo, hi):
        """The same designs for the sea states [lo, hi) only (shallow copy; used to shard one farm's load cases)."""
        import copy
        s = copy.copy(self)
        s.zeta = np.ascontiguousarray(self.zeta[lo:hi])
        s.beta = np.ascontiguousarray(self.beta[lo:hi])
        return s

    # ------------------------------------------------------------------ solve
    def upload(self, ctx):
        ctx.upload_designs_raw(self.off, self.strips, self.M0, self.B0, self.C0, self.nw, self.MBw, self.cmoff, self.cm)
        # the dynamic-pressure scale of the wave kinematics is Member.computeWaveKinematics' own default, not the site's
        # density (see GeometrySweep.upload)
        ctx.upload_cases(self.w, self.k, self.depth, WAVE_RHO, WAVE_G, self.zeta, self.beta)
        self._upload_bem(ctx)

    def solve(self, ctx, upload=True):
        """One launch over every (design, case) of this sweep; results stay on the device until fetched."""
        if upload:
            self.upload(ctx)
        ctx.solve_dynamics_device(self.nIter, self.tol, self.XiStart)
        return ctx.last_kernel_ms()

    def run_stats(self, ctx, want_psd=False):
        """Solve and return only the response statistics (std [nD,nC,6], optional PSD) + niter/flags:
        ~60 B per (design, case) cross the bus instead of 19 KB (raft_fowt.py:2310-2357)."""
        self.solve(ctx)
        dw = float(self.w[1] - self.w[0]) if self.nw > 1 else float(self.w[0])
        std, psd = ctx.motion_stats(dw, want_psd)
        r = ctx.fetch_results(want_Xi=False)
        return {"std": std, "psd": psd, "niter": r["niter"], "flags": r["flags"]}

    def run_channels(self, ctx, L, Gw=None, want_psd=False):
        """Solve and return the statistics of general linear output channels of every (design, case) --
        y_c = sum_p (i w)^p L[.,c,p,:] . Xi + Gw[.,c,:,w] . Xi (raftx_channel_stats_poly): platform motions, nacelle
        accelerations, tower-base bending moment (raft_amd.dropin.tower_base_rows) ... what an optimiser's constraints
        consume (raft/omdao_raft.py:840-876), a few hundred bytes per (design, case) instead of the responses."""
        self.solve(ctx)
        dw = float(self.w[1] - self.w[0]) if self.nw > 1 else float(self.w[0])
        std, psd = ctx.channel_stats_poly(L, dw, Gw=Gw, want_psd=want_psd)
        r = ctx.fetch_results(want_Xi=False)
        return {"std": std, "psd": psd, "niter": r["niter"], "flags": r["flags"]}

    def run_farm(self, ctx, n_unit, Cc=None, Mc=None, Bc=None):
        """Arrays: consecutive groups of ``n_unit`` designs are the units of one farm (raft_model.py:1164-1236).
        Two launches: the per-unit fixed points, then the coupled 6N x 6N solves fed from the resident Z / F_wave.
        Returns Xi [nGroup,nCase,nHead,6*n_unit,nw] plus per-unit niter/flags."""
        from ._abi import WANT_FWAVE, WANT_Z, WANT_BDRAG
        self.upload(ctx)
        # units with constant M, B, C: the coupled solve assembles their 6 x 6 impedances itself from the matrices + the
        # exported B_drag, so the fixed points run as the LEAN kernel with the excitation export and no Z leaves them;
        # frequency-dependent M(w), B(w) (turbine aerodynamics, BEM) export Z from the full-featured kernel as before
        lean = getattr(self, "MBw", None) is None
        ctx.solve_dynamics_device(self.nIter, self.tol, self.XiStart, want_mask=(WANT_BDRAG if lean else WANT_Z) | WANT_FWAVE)
        t_units = ctx.last_kernel_ms()
        Xi = ctx.solve_system_resident(n_unit, Mc, Bc, Cc)
        r = ctx.fetch_results(want_Xi=False)
        return {"Xi": Xi, "niter": r["niter"], "flags": r["flags"], "kernel_ms": (t_units, ctx.last_kernel_ms())}

    def run_second_order(self, ctx, qtf_tables, Mstruc, w2, k2, S0, rho_water=1025.0, kay=None):
        """potSecOrder == 1 for a whole batch (single wave heading): first-order fixed point, slender-body QTFs from the
        converged motions, second-order force, restarted fixed point (raft_model.py:1108-1131) -- five launches in total,
        QTFs never leave the device.  qtf_tables: one raft_amd.qtf.QtfTable per design; Mstruc [nD,6,6];
        S0 [nCase,nw] wave spectra; kay: optional list (per design) of HOST Kim & Yue tables for the sea states' heading
        (default: built on the device, raftx_qtf_kay, when the tables carry MacCamy-Fuchs members),
        [nCase][nw2,nw2,6] each.  Returns Xi, niter (both stages), flags, Fhydro_2nd [nD,nCase,6,nw]."""
        from . import waves
        if self.n_head != 1:
            raise ValueError("run_second_order handles single-heading sea states")
        nD, nC, nw, n2 = self.n_design, self.n_case, self.nw, len(w2)
        self.upload(ctx)
        ctx.set_linearisation_point(None, keep_last=True)
        ctx.solve_dynamics_device(self.nIter, self.tol, self.XiStart)
        r1 = ctx.fetch_results(want_Xi=True)
        XiLast = ctx.fetch_linearisation_point()
        # motion RAOs on the second-order grid (helpers.py:762-784, raft_fowt.py:2022-2024): on the device straight from
        # the resident responses; the CPU oracle (tests) takes the host loop
        Xi2 = None
        if not ctx.rlib.is_device:
            Xi2 = np.zeros((nD * nC, 6, n2), dtype=complex)
            for d in range(nD):
                for c in range(nC):
                    rao = waves.get_rao(r1["Xi"][d, c, 0], self.zeta[c, 0])
                    for j in range(6):
                        Xi2[d * nC + c, j] = np.interp(w2, self.w, rao[j], left=0, right=0)
        tabs = [qtf_tables[d] for d in range(nD) for _ in range(nC)]
        beta = np.array([self.beta[c, 0] for _ in range(nD) for c in range(nC)])
        Ms = np.array([Mstruc[d] for d in range(nD) for _ in range(nC)])
        kt = None if kay is None else np.array([kay[d][c] for d in range(nD) for c in range(nC)])
        if kt is None and ctx.rlib.is_device and any(len(t.kay_geom) for t in tabs):
            ctx.qtf_kay(tabs, beta, w2, k2, self.depth, rho_water, self.g)        # Kim & Yue tables built on the device
        ctx.qtf_slender(tabs, Xi2, beta, w2, k2, self.depth, rho_water, self.g, Ms, kt, fetch=False)
        dw = float(self.w[1] - self.w[0])
        _, f2 = ctx.qtf_force(w2, self.w, dw, np.array([S0[c] for _ in range(nD) for c in range(nC)]), qtf=None,
                              n_set=nD * nC)
        F2 = f2.reshape(nD, nC, 6, nw)
        ok = (r1["flags"] & 1).astype(bool)                      # only converged pairs take the second stage
        F_extra = np.where(ok[:, :, None, None], F2, 0.0)[:, :, None].astype(complex)
        ctx.set_linearisation_point(XiLast, keep_last=False)
        ctx.solve_dynamics_device(max(self.nIter - 1, 0), self.tol, self.XiStart, F_extra=F_extra)
        r2 = ctx.fetch_results(want_Xi=True)
        Xi = np.where(ok[:, :, None, None, None], r2["Xi"], r1["Xi"])
        niter = np.where(ok, r1["niter"] + r2["niter"], r1["niter"])
        flags = np.where(ok, r2["flags"], r1["flags"])
        return {"Xi": Xi, "niter": niter, "flags": flags, "Fhydro_2nd": np.where(ok[:, :, None, None], F2, 0.0)}

    def run(self, ctx):
        self.solve(ctx)
        r = ctx.fetch_results(want_Xi=True)
        return {"Xi": r["Xi"], "niter": r["niter"], "flags": r["flags"]}


class GeometrySweep(Sweep):
    """A sweep whose designs are MEMBER DESCRIPTIONS (raft_amd/geometry.py DesignTables): ``upload`` generates the
    strip tables, MacCamy-Fuchs tables, Morison added mass, hydrostatics and member inertia on the device
    (raftx_build_designs) instead of uploading packed strips.  M_extra / C_extra [nD,6,6] carry what is not geometry
    (rotor-nacelle assembly, point inertias, mooring stiffness); ``add_mask`` selects what the device adds to them.
    Everything else -- sharding by design, solve, statistics, farm and second-order paths -- is inherited."""

    def __init__(self, tables, M_extra, B0, C_extra, w, k, depth, zeta, beta, nIter, XiStart, tol=0.01, pose=None,
                 add_mask=7, MBw=None, rho=1025.0, g=9.81):
        self.tables = tables
        nD = tables.n_design
        self.off = np.zeros(nD + 1, dtype=np.int64)            # filled by upload() (the device counts the strips)
        self.strips = np.zeros((0, NFIELD))
        self.M0, self.B0, self.C0 = (np.ascontiguousarray(a, dtype=np.float64).reshape(nD, 6, 6) for a in (M_extra, B0, C_extra))
        self.w = np.ascontiguousarray(w, dtype=np.float64)
        self.k = np.ascontiguousarray(k, dtype=np.float64)
        self.depth, self.rho, self.g = float(depth), float(rho), float(g)
        zeta = np.asarray(zeta, dtype=np.float64)
        beta = np.asarray(beta, dtype=np.float64)
        if zeta.ndim == 2:
            zeta, beta = zeta[None], beta[None]
        self.zeta, self.beta = np.ascontiguousarray(zeta), np.ascontiguousarray(beta)
        self.nIter, self.XiStart, self.tol = int(nIter), float(XiStart), float(tol)
        self.MBw = None if MBw is None else np.ascontiguousarray(MBw, dtype=np.float64)
        self.pose = None if pose is None else np.ascontiguousarray(pose, dtype=np.float64).reshape(nD, 6)
        self.add_mask = int(add_mask)
        self.cmoff = self.cm = None

    @property
    def n_design(self):
        return self.tables.n_design

    def take(self, lo, hi):
        return self._take_bem(GeometrySweep(self.tables.take(lo, hi), self.M0[lo:hi], self.B0[lo:hi], self.C0[lo:hi], self.w,
                                            self.k, self.depth, self.zeta, self.beta, self.nIter, self.XiStart, self.tol,
                                            None if self.pose is None else self.pose[lo:hi], self.add_mask,
                                            None if self.MBw is None else self.MBw[lo:hi], self.rho, self.g), lo, hi)

    def run_crossing(self, ctx, n_chunk=0, n_worker=0, want_Xi=False, Xi_out=None):
        """The whole boundary crossing in ONE library call (raftx_sweep_stats): descriptors in, motion statistics +
        iteration counts (+ responses) out, with upload / generation / solve / download of consecutive design blocks
        overlapped on the library's internal streams.  Nothing stays resident on ``ctx``."""
        self._crossing_supported()
        t = self.tables
        r = ctx.sweep_stats(t, self.M0, self.B0, self.C0, self.w, self.k, self.depth, self.zeta, self.beta, self.nIter,
                            self.tol, self.XiStart, pose=self.pose, rho=self.rho, g=self.g, add_mask=self.add_mask,
                            n_chunk=n_chunk, n_worker=n_worker, want_Xi=want_Xi, Xi_out=Xi_out)
        self.off = r["strip_off"]
        return r

    def prepare_crossing(self, ctx, slot, n_chunk=0, want_Xi=False, Xi_out=None):
        """First stage of the streamed form of ``run_crossing`` for back-to-back batches: enqueue this batch's descriptor
        upload and member pass on ``slot`` (0 .. 3) and return a handle (raftx_sweep_prepare)."""
        self._crossing_supported()
        return ctx.sweep_prepare(slot, self.tables, self.M0, self.B0, self.C0, self.w, self.k, self.depth, self.zeta, self.beta,
                                 self.nIter, self.tol, self.XiStart, pose=self.pose, rho=self.rho, g=self.g, add_mask=self.add_mask,
                                 n_chunk=n_chunk, want_Xi=want_Xi, Xi_out=Xi_out)

    def launch_crossing(self, ctx, handle):
        """Second stage: table generation, fused fixed point and statistics of a prepared batch (raftx_sweep_launch).  With
        launch(i+1), prepare(i+2), wait(i) per step a long sweep keeps three batches in flight and the fused kernels of
        consecutive batches follow each other without a gap."""
        return ctx.sweep_launch(handle)

    def submit_crossing(self, ctx, slot, n_chunk=0, want_Xi=False, Xi_out=None):
        """prepare + launch in one call (raftx_sweep_submit); ``wait_crossing`` collects the results."""
        return self.launch_crossing(ctx, self.prepare_crossing(ctx, slot, n_chunk=n_chunk, want_Xi=want_Xi, Xi_out=Xi_out))

    def _crossing_supported(self):
        """raftx_sweep_stats / raftx_sweep_submit carry neither frequency-dependent matrices nor potential-flow excitation
        (include/raftx.h): a sweep that has them must go through upload() + solve(), never lose them silently."""
        if self.MBw is not None or getattr(self, "bem", None) is not None:
            raise ValueError("the one-call crossing does not carry MBw / BEM excitation: use upload() + solve() (+ run_stats)")

    def wait_crossing(self, ctx, handle):
        r = ctx.sweep_wait(handle)
        self.off = r["strip_off"]
        return r

    def upload(self, ctx):
        t = self.tables
        self.off = ctx.build_designs(t.member_off, t.members, t.station_off, t.stations, self.M0, self.B0, self.C0, self.nw,
                                     pose=self.pose, rho=self.rho, g=self.g, k=self.k, add_mask=self.add_mask, MBw=self.MBw,
                                     cap_off=t.cap_off, caps=t.caps)
        # the dynamic-pressure scale of the wave kinematics is NOT the site's density: FOWT.calcHydroExcitation never
        # forwards rho / g to Member.calcHydroExcitation (raft_fowt.py:1857), whose defaults are 1025 / 9.81
        # (raft_member.py:1940); self.rho / self.g are the site values the strip constants and statics are built with
        ctx.upload_cases(self.w, self.k, self.depth, WAVE_RHO, WAVE_G, self.zeta, self.beta)
        self._upload_bem(ctx)


# ---------------------------------------------------------------------- multi-GPU drivers
# ``comm`` is a raft_amd.comm communicator (RcclComm on the GPUs, HostComm in CPU tests / rehearsals): rank, world,
# broadcast_arrays, gather_rows, gather_xi, reduce_sum.  No collective is issued while a kernel runs.
def _counts(n, world):
    return np.array([shard_bounds(n, r, world)[1] - shard_bounds(n, r, world)[0] for r in range(world)], dtype=np.int64)


def broadcast_cases(cases, comm):
    """Rank 0 broadcasts the shared sea-state tables (dict of small float64 arrays + scalars: w, k, zeta, beta, depth)."""
    if comm is None or comm.world == 1:
        return cases
    if comm.rank == 0:
        cases = {k_: (np.ascontiguousarray(v, dtype=np.float64) if isinstance(v, (np.ndarray, list, tuple)) else v)
                 for k_, v in cases.items()}
    return comm.broadcast_arrays(cases if comm.rank == 0 else None)


def run_sharded(sweep, ctx, comm=None, gather=True):
    """Every rank solves its block of designs; rank 0 gets the assembled results
    ({"Xi","niter","flags"}), the others their local block only.  The responses travel from the HBM buffers they
    were solved into (comm.gather_xi), the small integer arrays as rows."""
    if comm is None or comm.world == 1:
        return sweep.run(ctx)
    rank, world = comm.rank, comm.world
    sub = sweep.shard(rank, world)
    sub.solve(ctx)
    r = ctx.fetch_results(want_Xi=not gather or rank != 0)
    local = {"Xi": r["Xi"], "niter": r["niter"], "flags": r["flags"]}
    if not gather:
        return local
    cd = _counts(sweep.n_design, world)
    Xi = comm.gather_xi(ctx, cd * sweep.n_case)
    ni = comm.gather_rows(local["niter"], cd)
    fl = comm.gather_rows(local["flags"], cd)
    if rank != 0:
        return local
    return {"Xi": Xi.reshape((sweep.n_design, sweep.n_case) + Xi.shape[1:]), "niter": ni, "flags": fl}


def shard_fingerprint(sweep, lo, hi):
    """sha256 over everything the statistics of designs [lo, hi) depend on: their tables (packed strips or member
    descriptions), matrices, the sea states and the solver settings.  A checkpointed shard is reused only if it matches."""
    import hashlib
    h = hashlib.sha256()

    def add(a):
        if a is None:
            h.update(b"<none>")
            return
        a = np.ascontiguousarray(a)
        h.update(str(a.dtype).encode() + str(a.shape).encode())
        h.update(a.tobytes())

    sub = sweep.take(lo, hi)
    t = getattr(sub, "tables", None)
    if t is not None:                                            # GeometrySweep: member descriptions
        for name in ("member_off", "members", "station_off", "stations", "cap_off", "caps"):
            add(getattr(t, name, None))
        add(np.array([sub.add_mask]))
        add(sub.pose)
        prog = getattr(sub, "program", None)
        if prog is not None:                                         # VariantSweep: the base unit, the edit program, the parameter rows
            for a in (prog.base.members, prog.base.stations, prog.base.caps, prog.end_coef, prog.end_edit, prog.head_cs,
                      prog.dia_coef, prog.dia_edit, sub.params):
                add(a)
    else:
        add(sub.off)
        add(sub.strips)
        add(getattr(sub, "cmoff", None))
        add(getattr(sub, "cm", None))
    for a in (sub.M0, sub.B0, sub.C0, sub.MBw, sub.w, sub.k, sub.zeta, sub.beta):
        add(a)
    b = getattr(sub, "bem", None)
    if b is not None:
        for k_ in ("heads", "X", "hadj", "xy"):
            add(b[k_])
    add(np.array([sub.depth, sub.rho, sub.g, sub.tol, sub.XiStart, float(sub.nIter), float(sub.n_case), float(lo), float(hi)]))
    return h.hexdigest()


def run_stats_sharded(sweep, ctx, comm=None, checkpoint_dir=None, shards_per_rank=1):
    """The optimiser-style exchange: every rank solves its designs and only the motion statistics (48 B per design-case)
    and iteration counts are gathered onto rank 0.

    checkpoint_dir: the sweep is idempotent, so it is made resumable the simple way: the designs are
    cut into ``world * shards_per_rank`` contiguous shards, a rank writes every shard it finishes as
    ``shard_<id>_of_<n>.npz`` (atomically: temp file + rename) and, on a re-run with the same directory, loads the shards
    that are already there instead of solving them -- whichever rank wrote them.  The gathered result is the same
    arrays either way."""
    world = 1 if comm is None else comm.world
    rank = 0 if comm is None else comm.rank
    keys = ("std", "niter", "flags")
    if checkpoint_dir is None:
        if world == 1:
            return sweep.run_stats(ctx)
        local = sweep.shard(rank, world).run_stats(ctx)
        counts = _counts(sweep.n_design, world)
    else:
        import os
        os.makedirs(checkpoint_dir, exist_ok=True)
        n_shard = world * max(int(shards_per_rank), 1)
        first = lambda r: r * n_shard // world                      # rank r owns the shards first(r) .. first(r + 1) - 1
        rows = lambda r: shard_bounds(sweep.n_design, first(r + 1) - 1, n_shard)[1] - shard_bounds(sweep.n_design, first(r), n_shard)[0]
        parts = []
        for sid in range(first(rank), first(rank + 1)):
            path = os.path.join(checkpoint_dir, "shard_%05d_of_%05d.npz" % (sid, n_shard))
            lo, hi = shard_bounds(sweep.n_design, sid, n_shard)
            got = None
            fp = shard_fingerprint(sweep, lo, hi)
            if os.path.exists(path):
                with np.load(path) as z:                         # a shard of another sweep in the same directory is recomputed
                    if int(z["lo"]) == lo and int(z["hi"]) == hi and "fp" in z.files and str(z["fp"]) == fp:
                        got = {k_: z[k_] for k_ in keys}
            if got is None:
                if hi > lo:
                    got = sweep.take(lo, hi).run_stats(ctx)
                else:
                    got = {"std": np.zeros((0, sweep.n_case, 6)), "niter": np.zeros((0, sweep.n_case), np.int32),
                           "flags": np.zeros((0, sweep.n_case), np.int32)}
                tmp = path + ".tmp.%d" % os.getpid()
                with open(tmp, "wb") as f:
                    np.savez(f, lo=lo, hi=hi, fp=np.array(fp), **{k_: got[k_] for k_ in keys})
                os.replace(tmp, path)
            parts.append(got)
        local = {k_: np.concatenate([p[k_] for p in parts], axis=0) for k_ in keys}
        if world == 1:
            return local
        counts = np.array([rows(r) for r in range(world)], dtype=np.int64)
    out = {k_: comm.gather_rows(local[k_], counts) for k_ in keys}
    return out if rank == 0 else local


def run_qtf_sharded(qtf_fn, tables, Xi, beta, w2, k2, depth, rho, g, Mstruc, kay=None, comm=None):
    """Slender-body QTFs of many (table, motion, heading) sets, sharded by set over the ranks (sets are independent:
    no collective while computing); rank 0 gets [nSet,nw2,nw2,6], the others their own block.
    qtf_fn = ctx.qtf_slender of the rank's context (tests pass the numpy oracle)."""
    n = len(tables)
    Xi, beta, Mstruc = np.asarray(Xi), np.asarray(beta), np.asarray(Mstruc)
    if comm is None or comm.world == 1:
        return qtf_fn(tables, Xi, beta, w2, k2, depth, rho, g, Mstruc, kay)
    rank, world = comm.rank, comm.world
    lo, hi = shard_bounds(n, rank, world)
    local = qtf_fn(tables[lo:hi], Xi[lo:hi], beta[lo:hi], w2, k2, depth, rho, g, Mstruc[lo:hi],
                   None if kay is None else np.asarray(kay)[lo:hi])
    full = comm.gather_rows(local, _counts(n, world))
    return full if rank == 0 else local


def run_qtf_rows_sharded(qtf_fn, tables, Xi, beta, w2, k2, depth, rho, g, Mstruc, kay=None, comm=None):
    """ONE QTF (or a few) shared by all ranks -- SURVEY.md 8e C5: every rank computes the INTERLEAVED rows
    w1 = w2[rank::world] of every set (row i1 holds nw2 - i1 pairs, so contiguous blocks would be unbalanced) with
    ``qtf_fn(..., rows=(rank, world))`` (ctx.qtf_slender -> raftx_qtf_slender_rows), which returns zeros elsewhere;
    the partial matrices are SUMMED onto rank 0 (one reduce of nSet*nw2^2*6 complex: 3.8 MB for the 200-point grid).
    Rank 0 gets the full [nSet,nw2,nw2,6]; the others their partial."""
    if comm is None or comm.world == 1:
        return qtf_fn(tables, Xi, beta, w2, k2, depth, rho, g, Mstruc, kay)
    part = np.ascontiguousarray(qtf_fn(tables, Xi, beta, w2, k2, depth, rho, g, Mstruc, kay, rows=(comm.rank, comm.world)))
    total = comm.reduce_sum(part)
    return total if comm.rank == 0 else part


def run_farm_sharded(sweep, ctx, n_unit, Cc=None, Mc=None, Bc=None, comm=None):
    """ONE farm (or a few) and many sea states -- SURVEY.md 8e C4: the CASES are block-partitioned over the ranks
    (every rank holds all units' tables: 54 KB for four VolturnUS-S), each rank runs ``Sweep.run_farm`` on its sea
    states, and the responses are gathered along the case axis onto rank 0.  No collective while solving."""
    if comm is None or comm.world == 1:
        return sweep.run_farm(ctx, n_unit, Cc=Cc, Mc=Mc, Bc=Bc)
    rank, world = comm.rank, comm.world
    lo, hi = shard_bounds(sweep.n_case, rank, world)
    local = sweep.take_cases(lo, hi).run_farm(ctx, n_unit, Cc=Cc, Mc=Mc, Bc=Bc)
    counts = _counts(sweep.n_case, world)
    out = {}
    for key, v in local.items():                         # case axis is axis 1 of every array result
        if np.ndim(v) < 2:                               # kernel_ms etc.: per-rank scalars
            out[key] = v
            continue
        g = comm.gather_rows(np.ascontiguousarray(np.moveaxis(v, 1, 0)), counts)
        out[key] = None if g is None else np.moveaxis(g, 0, 1)
    return out if rank == 0 else local


def run_flex_sharded(flex_sweep, ctx, comm=None):
    """A sweep of units with flexible members (raft_amd/flex.py FlexSweep): the UNITS are block-partitioned over the ranks, every
    rank runs the whole fixed point of its units x all sea states on its GPU (raftx_flex_solve: no collective while solving),
    and rank 0 gathers responses, iteration counts, flags and B_drag along the unit axis.  Rank 0 returns the assembled
    dict, the others their local block."""
    from .flex import FlexSweep
    if comm is None or comm.world == 1:
        return flex_sweep.run(ctx)
    rank, world = comm.rank, comm.world
    n = len(flex_sweep.units)
    lo, hi = shard_bounds(n, rank, world)
    counts = _counts(n, world)
    local = None
    if hi > lo:
        sub = FlexSweep(flex_sweep.units[lo:hi], flex_sweep.w, flex_sweep.k, flex_sweep.depth, flex_sweep.zeta, flex_sweep.beta,
                        flex_sweep.nIter, flex_sweep.XiStart, flex_sweep.tol)
        try:
            local = sub.run(ctx)                     # copy=True: the results own their memory ...
        finally:
            sub.release(ctx)                         # ... so the throw-away sweep's page-locked arrays go back at once
    nC, nH, nw, nd = flex_sweep.zeta.shape[0], flex_sweep.zeta.shape[1], len(flex_sweep.w), flex_sweep.n
    empty = {"Xi": np.zeros((0, nC, nH, nd, nw), dtype=complex), "niter": np.zeros((0, nC), dtype=np.int32),
             "flags": np.zeros((0, nC), dtype=np.int32), "B_drag": np.zeros((0, nC, nd, nd))}
    out = {}
    for key in ("Xi", "niter", "flags", "B_drag"):
        out[key] = comm.gather_rows(np.ascontiguousarray((local or empty)[key]), counts)
    if rank != 0:
        return local
    out["kernel_ms"] = local["kernel_ms"] if local else (0.0, 0.0)
    return out


class Pipeline:
    """Host-buffer boundary with copies overlapped with compute.  ``n_workers`` raftx contexts (= HIP streams with their
    own device buffers and memory pools) live as long as the Pipeline; ``run`` cuts the designs of a sweep into blocks
    that the workers -- one Python thread per context -- take round-robin: upload / generate, solve, download.  ctypes
    releases the GIL inside every C call, so while one worker's stream runs the fused kernel the other workers' H2D /
    D2H copies are in flight.  Results come back in design order, bit-identical to ``sweep.run``."""

    def __init__(self, lib, n_workers=2, device_id=0):
        self.ctxs = [lib.context(device_id) for _ in range(max(1, int(n_workers)))]

    def close(self):
        self._pinned_out = None
        for c in self.ctxs:
            c.close()
        self.ctxs = []

    def run(self, sweep, n_chunks=None, fetch="Xi", pinned=False):
        """fetch: "Xi" (full responses) or "stats" (motion std + iteration counts only).  pinned=True: the responses are
        downloaded into a page-locked array of the first context (faster D2H); it is recycled by the next pinned run
        and released by close(), so copy what must outlive them."""
        import threading
        nD, nW = sweep.n_design, len(self.ctxs)
        n_chunks = max(1, min(int(n_chunks or 4 * nW), nD))
        bounds = [shard_bounds(nD, i, n_chunks) for i in range(n_chunks)]
        parts = [None] * n_chunks
        errors = []
        # full responses land straight in their block of ONE preallocated array (no concatenation of 19 KB x nD x nC)
        Xi_all = None
        if fetch != "stats":
            shape = (nD, sweep.n_case, sweep.n_head, 6, sweep.nw)
            if pinned:
                old = getattr(self, "_pinned_out", None)
                if old is None or old.shape != shape:
                    if old is not None:
                        self.ctxs[0].free_pinned(old)
                    self._pinned_out = self.ctxs[0].pinned_empty(shape)
                Xi_all = self._pinned_out
            else:
                Xi_all = np.empty(shape, dtype=np.complex128)

        def worker(wid):
            ctx = self.ctxs[wid]
            try:
                for i in range(wid, n_chunks, nW):
                    lo, hi = bounds[i]
                    sub = sweep.take(lo, hi)
                    if fetch == "stats":
                        parts[i] = sub.run_stats(ctx)
                    else:
                        sub.solve(ctx)
                        r = ctx.fetch_results(Xi_out=Xi_all[lo:hi])
                        parts[i] = {"niter": r["niter"], "flags": r["flags"]}
            except Exception as e:                               # surfaced after the join
                errors.append(e)

        threads = [threading.Thread(target=worker, args=(w,)) for w in range(nW)]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        if errors:
            raise errors[0]
        keys = [k_ for k_ in parts[0] if parts[0][k_] is not None and np.ndim(parts[0][k_]) >= 1]
        out = {k_: np.concatenate([p[k_] for p in parts], axis=0) for k_ in keys}
        if Xi_all is not None:
            out["Xi"] = Xi_all
        return out


def run_pipelined(sweep, lib, n_chunks=8, n_workers=2, device_id=0, fetch="Xi"):
    """One-shot form of Pipeline (contexts created and destroyed inside: cold memory pools)."""
    pipe = Pipeline(lib, n_workers, device_id)
    try:
        return pipe.run(sweep, n_chunks, fetch)
    finally:
        pipe.close()


class VariantSweep(GeometrySweep):
    """A sweep whose designs are PARAMETRIC VARIANTS of one base unit (raft_amd/geometry.py VariantProgram;
    raft/parametersweep.py:39-87): per candidate only its parameter values cross the bus, the member / station / cap
    descriptors are written on the device (raftx_sweep_prepare_variants -> k_geom_expand) and everything after that is
    the GeometrySweep path.  ``params`` [nD,nParam]; M_extra / B0 / C_extra as GeometrySweep.

    ``set_params`` swaps in the next batch's candidates (same batch size: the library's slots stay configured).  The
    crossing calls (prepare / submit / wait, run_crossing) take the device path; ``upload`` / ``solve`` / ``run`` -- the
    resident forms the checks use -- expand the descriptors through the library once (ctx.expand_variants) and hand them
    to raftx_build_designs, so both routes see the same rows."""

    def __init__(self, program, params, M_extra, B0, C_extra, w, k, depth, zeta, beta, nIter, XiStart, tol=0.01, pose=None,
                 add_mask=7, rho=1025.0, g=9.81):
        self.program = program
        self.params = np.ascontiguousarray(params, dtype=np.float64).reshape(-1, program.n_param)
        self._tables = None
        self._installed = None
        super().__init__(_VariantTables(self), M_extra, B0, C_extra, w, k, depth, zeta, beta, nIter, XiStart, tol=tol, pose=pose,
                         add_mask=add_mask, rho=rho, g=g)

    @property
    def n_design(self):
        return self.params.shape[0]

    def set_params(self, params):
        params = np.ascontiguousarray(params, dtype=np.float64).reshape(-1, self.program.n_param)
        if params.shape != self.params.shape:
            raise ValueError("set_params: a batch of the same size is expected (%r, got %r)" % (self.params.shape, params.shape))
        self.params = params
        self._tables = None

    def _install(self, ctx):
        if self._installed is not ctx or getattr(ctx, "_vprog_owner", None) is not self.program:
            ctx.variant_program(self.program)
            ctx._vprog_owner = self.program
            self._installed = ctx

    def expanded_tables(self, ctx):
        """DesignTables of the current params, expanded by the library on ``ctx`` (cached until set_params)."""
        if self._tables is None:
            self._install(ctx)
            self._tables = self.program.tables(ctx.expand_variants(self.params), self.n_design)
        return self._tables

    def take(self, lo, hi):
        sub = VariantSweep(self.program, self.params[lo:hi], self.M0[lo:hi], self.B0[lo:hi], self.C0[lo:hi], self.w, self.k, self.depth,
                           self.zeta, self.beta, self.nIter, self.XiStart, self.tol, None if self.pose is None else self.pose[lo:hi],
                           self.add_mask, self.rho, self.g)
        return self._take_bem(sub, lo, hi)

    def prepare_crossing(self, ctx, slot, n_chunk=0, want_Xi=False, Xi_out=None):
        self._crossing_supported()
        self._install(ctx)
        return ctx.sweep_prepare_variants(slot, self.params, self.M0, self.B0, self.C0, self.w, self.k, self.depth, self.zeta, self.beta,
                                          self.nIter, self.tol, self.XiStart, pose=self.pose, rho=self.rho, g=self.g,
                                          add_mask=self.add_mask, n_chunk=n_chunk, want_Xi=want_Xi, Xi_out=Xi_out)

    def run_crossing(self, ctx, n_chunk=0, n_worker=0, want_Xi=False, Xi_out=None, slot=0):
        """One isolated crossing: prepare + launch + wait on ``slot``."""
        return self.wait_crossing(ctx, self.submit_crossing(ctx, slot, n_chunk=n_chunk, want_Xi=want_Xi, Xi_out=Xi_out))

    def upload(self, ctx):
        self.tables = self.expanded_tables(ctx)
        try:
            super().upload(ctx)
        finally:
            self.tables = _VariantTables(self)


class _VariantTables:
    """What GeometrySweep asks its ``tables`` for when the descriptors live on the device: the batch size only."""

    def __init__(self, sweep):
        self._sweep = sweep

    @property
    def n_design(self):
        return self._sweep.params.shape[0]
