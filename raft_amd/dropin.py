"""Drop-in replacements for the reference's hot-path methods.

Same names, arguments, return values and side effects as
    raft/raft_model.py:966    Model.solveDynamics(case, tol=0.01, conv_plot=0, RAO_plot=0, display=0)
    raft/raft_fowt.py:1732    FOWT.calcHydroExcitation(case, memberList=[])
    raft/raft_fowt.py:1891    FOWT.calcHydroLinearization(Xi)
    raft/raft_fowt.py:1940    FOWT.calcDragExcitation(ih)
but the strip sweeps, the drag-linearisation fixed point and the per-bin
solves run in libraftx_hip.so (hand-written gfx950 kernels) through the C-ABI
of include/raftx.h.  ``install()`` monkey-patches a loaded ``raft`` package;
the functions also work on any duck-typed stand-ins exposing the attributes
read here (that is how the GPU-box tests run without /root/reference).

    raft/raft_fowt.py:1988    FOWT.calcQTF_slenderBody(waveHeadInd, Xi0=None, verbose=False, iCase=None, iWT=None)
    raft/raft_fowt.py:2158    FOWT.calcHydroForce_2ndOrd(beta, S0, iCase=None, iWT=None, interpMode='qtf')
are mirrored too (internal slender-body QTFs, potSecOrder == 1, incl. the re-entry of the drag iteration with the
second-order force, raft_model.py:1108-1131).

On the device path too: units with more than 6 reduced DOFs (flexible members; Engine._solve_general, raftx_flex_solve),
potential-flow coefficients on either kind of unit, submerged rotors on any unit of an array, arrays with a shared
lumped-mass mooring (array-level moorMod == 2, raft_model.py:1173-1182).  A unit's own moorMod == 2 mooring (per-iteration
line damping from the mooring model, raft_model.py:1019-1030,1069-1072) is honoured by stepping the fixed point one device
launch per iteration from an explicit linearisation point (Engine._solve_stepped; for units with more than 6 reduced DOFs
the stepped branch of Engine._solve_general, raftx_flex_start).
Per-member intermediates (mem.u, mem.ud, mem.pDyn, mem.F_hydro_iner, mem.Bmat, mem.F_exc_drag) are never needed by the
replaced methods; ``install(materialise_members=True)`` fills them from a device export for un-replaced callers.

Not covered (raises UnsupportedFOWT, never falls back silently): arrays of units with more than 6 reduced DOFs (upstream
cannot build them: an array's per-unit design drops `joints`, raft_model.py:113-137, raft_fowt.py:205), second-order loads on
such units (stubs upstream), moorMod == 2 together with internal QTFs, a dry unit with moorMod == 2.
"""
import numpy as np

from . import waves
from .rigid import translate_matrix_6to6
from .strips import pack_fowt, pack_fowt_nodes, pack_fingerprint, UnsupportedFOWT
from . import backend
from .hostblas import few_threads


def _sum_rotors(A):
    """np.sum(A, axis=3) of the per-rotor matrices [n,n,nw,nrotors] (raft_model.py:1005-1010): one rotor -> its slice (the
    sum over an axis of length one copies 7 MB per matrix at 150 DOFs to return the same numbers)."""
    A = np.asarray(A)
    return A[:, :, :, 0] if A.shape[3] == 1 else np.sum(A, axis=3)


def _nonzero(a):
    """np.any(a) for the large real tables of a call (rotor / potential-flow matrices [n,n,nw]: 7 MB at 150 DOFs, all zeros
    for a parked turbine -- np.any scans them at 1 GB/s, 0.75 ms each on the GPU box's host): a BLAS dot product answers
    "something is there" (NaN included) at memory speed; if it is 0 -- all zeros, or values whose squares underflow -- the
    bit patterns decide (-0.0 counts as something: adding it changes nothing)."""
    a = np.asarray(a)
    if a.dtype != np.float64 or a.size < 65536 or not a.flags.c_contiguous:
        return bool(np.any(a))
    f = a.reshape(-1)
    with few_threads():
        d = f.dot(f)
    if d != 0.0:
        return True
    return bool(f.view(np.uint64).max() != 0)


def _real_times_complex(T, X):
    """T [m,n] real times X [...,n,nw] complex as ONE real matrix product on the interleaved (re, im) view of X: `T @ X` with
    mixed types takes NumPy's slow path (no BLAS); T (re, im) = (T re, T im) is the same arithmetic as a dgemm."""
    Xc = np.ascontiguousarray(X, dtype=complex)
    with few_threads():
        Y = np.matmul(np.asarray(T, dtype=float), Xc.view(np.float64).reshape(Xc.shape[:-1] + (2 * Xc.shape[-1],)))
    return np.ascontiguousarray(Y).view(np.complex128)


def _quasi_static_tension_rows(fowt):
    """(J_moor [2 nLines, 6], T_moor [2 nLines]) of a unit's MoorPy system for the quasi-static tension block of
    FOWT.saveTurbineOutputs (raft_fowt.py:2356-2399), or (None, None) without one.  Host calls into the third-party system, as
    upstream makes them; line dynamics (moorMod != 0) is MoorPy's own frequency-domain line solver and stays upstream."""
    if not getattr(fowt, "ms", None):
        return None, None
    if getattr(fowt, "moorMod", 0) != 0:
        raise UnsupportedFOWT("saveTurbineOutputs: line-dynamics mooring tensions (moorMod != 0, raft_fowt.py:2374-2387: "
                              "MoorPy's dynamicSolve per line) are not on the device path")
    try:                                                                     # composite lines -> subsystems, as :2358 does
        from moorpy.helpers import lines2ss
        fowt.ms = lines2ss(fowt.ms)
    except ImportError:                                                      # (a stand-in system without MoorPy: nothing to convert)
        pass
    _, J = fowt.ms.getCoupledStiffness(lines_only=True, tensions=True)       # :2363
    J = np.asarray(J, dtype=float)
    Tm = np.asarray(fowt.ms.getTensions(), dtype=float)                      # :2364, mean line-end tensions
    if J.shape != (2 * len(fowt.ms.lineList), 6) or Tm.shape != (J.shape[0],):
        raise UnsupportedFOWT("saveTurbineOutputs: the mooring system's tension Jacobian is not [2 nLines, 6]")
    return J, Tm


class Engine:
    """Binds the host mirror to one raftx context (default: the HIP library)."""

    def __init__(self, ctx=None, qtf_backend=None, materialise_members=False):
        self._ctx = ctx
        # materialise_members: also leave on every Member what the reference's per-member methods leave there -- u, ud,
        # pDyn (raft_member.py:1927-1937), F_hydro_iner (:1991), Bmat and F_exc_drag (:2117,2122) -- through the library's
        # per-strip export (raftx_strip_kinematics / raftx_strip_drag), so that un-replaced reference code that reads them
        # (Member.calcDragExcitation, plotting, post-processing) sees current arrays.  Off by default: the arrays are
        # ~1 MB per heading at C2 and the fused path never needs them.
        self.materialise_members = bool(materialise_members)
        # qtf_backend(tables, Xi, beta, w2, k2, depth, rho, g, Mstruc, kay) -> qtf [nSet,nw2,nw2,6]; default: the
        # device kernels through the C-ABI (tests inject the numpy oracle to exercise the host logic on CPU)
        self._qtf_backend = qtf_backend

    @property
    def ctx(self):
        if self._ctx is None:
            self._ctx = backend.default_context()
        return self._ctx

    # ------------------------------------------------------------------ per-member side effects, on request
    @staticmethod
    def _scatter(table, members, arrays, heading_axis):
        """Rows of the unit's strip table -> (member, strip) slots of per-member arrays.  arrays: list of (attribute name,
        per-strip array with the strip axis at ``heading_axis``, trailing shape after the strip axis)."""
        from .strips import F_MEM, F_IL
        rows_m = np.asarray(table.strips[:, F_MEM], dtype=int)
        rows_il = np.asarray(table.strips[:, F_IL], dtype=int)
        for im, mem in enumerate(members):
            sel = np.nonzero(rows_m == im)[0]
            for name, arr, lead in arrays:
                tgt = getattr(mem, name)
                if len(sel):
                    if lead:                                   # [nHead, strip, ...]
                        tgt[:, rows_il[sel]] = arr[:, sel]
                    else:                                      # [strip, ...]
                        tgt[rows_il[sel]] = arr[sel]

    def _no_materialise_general(self, fowt):
        """The per-strip export addresses the strips of ONE rigid table; units with more than 6 reduced DOFs keep a table per
        structural node -- asked to materialise their members, say so instead of leaving stale arrays behind."""
        if self.materialise_members and _general(fowt):
            raise UnsupportedFOWT("materialise_members is available for rigid 6-DOF units only (this unit has %d reduced DOFs)"
                                  % int(fowt.nDOF))

    def _materialise_kinematics(self, fowt, members, design=0):
        """mem.u, mem.ud [nWaves,ns,3,nw], mem.pDyn [nWaves,ns,nw] of every member of ``members`` (zeros for strips above
        the waterline, raft_member.py:1927-1937) from the device's per-strip export of resident design ``design``."""
        table = fowt._raftx_table
        nw, nH = fowt.nw, fowt.nWaves
        for mem in members:
            mem.u = np.zeros([nH, mem.ns, 3, nw], dtype=complex)
            mem.ud = np.zeros([nH, mem.ns, 3, nw], dtype=complex)
            mem.pDyn = np.zeros([nH, mem.ns, nw], dtype=complex)
        if len(table.strips):
            u, ud, p = self.ctx.strip_kinematics(design, len(table.strips))
            self._scatter(table, members, [("u", u, True), ("ud", ud, True), ("pDyn", p, True)], 1)

    def _materialise_drag(self, fowt, members, Xi, ih, design=0):
        """mem.Bmat [ns,3,3] of the linearisation about Xi [6,nw] and mem.F_exc_drag [ns,3,nw] = Bmat u[ih]
        (raft_member.py:2117,2122,2146) from the device's per-strip export."""
        table = fowt._raftx_table
        nw = fowt.nw
        for mem in members:
            if not hasattr(mem, "Bmat") or np.shape(mem.Bmat) != (mem.ns, 3, 3):
                mem.Bmat = np.zeros([mem.ns, 3, 3])
            if not hasattr(mem, "F_exc_drag") or np.shape(mem.F_exc_drag) != (mem.ns, 3, nw):
                mem.F_exc_drag = np.zeros([mem.ns, 3, nw], dtype=complex)
        if len(table.strips):
            B, F = self.ctx.strip_drag(design, len(table.strips), Xi, ih=ih)
            self._scatter(table, members, [("Bmat", B, False), ("F_exc_drag", F, False)], 0)

    # ------------------------------------------------------------------
    def _sea_state(self, fowt, case):
        nWaves, beta, S, zeta = waves.sea_state(case, fowt.w, fowt.dw)
        fowt.nWaves, fowt.beta, fowt.S, fowt.zeta = nWaves, beta, S, zeta

    def _check_supported(self, fowt):
        if _general(fowt):
            if any(rot.r3[2] < 0 for rot in getattr(fowt, "rotorList", [])):
                raise UnsupportedFOWT("submerged rotors on a unit with %d reduced DOFs" % fowt.nDOF)
            # more than 6 reduced DOFs (flexible members): strip theory node by node + the dense impedance solve.  Potential-flow
            # coefficients ride along as upstream lumps them: A_BEM, B_BEM in the first six DOFs of the reduced matrices
            # (raft_fowt.py:1479-1480), the excitation in the first six rows of the full-DOF vector (:1796-1849) reduced by T.
            # The second-order branches of such units are stubs upstream (calcQTF_slenderBody returns zeros for nDOF > 6)
            if int(getattr(fowt, "potSecOrder", 0)) != 0:
                raise UnsupportedFOWT("second-order loads on a unit with %d reduced DOFs" % fowt.nDOF)
            # a unit-level lumped-mass mooring (moorMod == 2) is lumped at the first six reduced DOFs (raft_model.py:1019-1030)
            # and re-linearised about every iterate (:1069-1072): _solve_general steps the fixed point, one launch per iteration

    # ------------------------------------------------------------------ units with more than 6 reduced DOFs
    def _node_units(self, fowt, members):
        """Strip tables per structural node + the rows of T that map the reduced DOFs onto each node."""
        T = np.asarray(fowt.T, dtype=float)
        # one packing per pose, not per load case (as pack_fingerprint for the rigid units): everything the node tables
        # read, the nodes they hang on and the reduction T
        fp = (pack_fingerprint(fowt, members), tuple(int(nd.id) for mem in members for nd in mem.nodeList), T.tobytes())
        cached = getattr(fowt, "_raftx_nodes_key", None)
        if cached is not None and cached == fp and getattr(fowt, "_raftx_nodes", None) is not None:
            return fowt._raftx_nodes
        rows, tables = pack_fowt_nodes(fowt, members)
        Tn = np.array([T[r:r + 6, :] for r in rows]).reshape(len(rows), 6, T.shape[1])
        fowt._raftx_nodes = (rows, tables, Tn)
        fowt._raftx_nodes_key = fp
        return fowt._raftx_nodes

    def _excitation_general(self, fowt, case, members):
        """raft_fowt.py:1853-1857,1886-1888 for nDOF > 6: per-node 6-vectors -> full-DOF rows -> T^T."""
        self._sea_state(fowt, case)
        nw, nFull, nDOF = fowt.nw, int(fowt.nFullDOF), int(fowt.nDOF)
        rows, tables, Tn = self._node_units(fowt, members)
        fowt.F_hydro_iner_fullDOF = np.zeros([fowt.nWaves, nFull, nw], dtype=complex)
        if tables:
            self._upload([fowt], fowt.zeta, fowt.beta, tables=tables)
            F = self.ctx.excitation()[:, 0]                  # [nNode, nWaves, 6, nw]
            for i, r in enumerate(rows):
                fowt.F_hydro_iner_fullDOF[:, r:r + 6, :] += F[i]
        T = np.asarray(fowt.T, dtype=float)
        fowt.F_hydro_iner = _real_times_complex(T.T, fowt.F_hydro_iner_fullDOF)                              # :1888
        fowt.F_BEM = np.zeros([fowt.nWaves, nDOF, nw], dtype=complex)
        fowt.F_BEM_fullDOF = np.zeros([fowt.nWaves, nFull, nw], dtype=complex)
        if getattr(fowt, "potMod", False) or int(getattr(fowt, "potModMaster", 0)) in (2, 3):
            # potential-flow excitation (:1796-1849), lumped at the first six full DOFs: on the device for the unit + sea
            # state resident (its node tables; the unit's coefficients ride on the first of them), reduced by T (:1887)
            if not tables:
                raise UnsupportedFOWT("potential-flow excitation of a unit with %d reduced DOFs and no wet strips" % nDOF)
            self._bem_excitation_units([fowt], n_pad=len(tables) - 1)
        fowt._raftx_fresh = True
        return None

    def _linearization_general(self, fowt, Xi):
        """raft_fowt.py:1891-1936 for nDOF > 6: node motions T_node Xi -> per-node 6 x 6 blocks and drag excitation ->
        T^T B_full T, T^T F_full."""
        rows, tables, Tn = fowt._raftx_nodes
        nDOF, nw, nH = int(fowt.nDOF), fowt.nw, fowt.nWaves
        B_red = np.zeros([nDOF, nDOF])
        F_red = np.zeros([nH, nDOF, nw], dtype=complex)
        if tables:
            self._upload([fowt], fowt.zeta, fowt.beta, tables=tables)     # no-op while these tables are resident
            nU = len(rows)
            T2 = Tn.reshape(nU * 6, nDOF)                                  # the nodes' rows of T, stacked: plain GEMMs
            XiN = _real_times_complex(T2, Xi).reshape(nU, 6, nw)
            B, F = self.ctx.linearize(XiN[:, None, :, :])                  # [nNode,1,6,6], [nNode,1,nWaves,6,nw]
            with few_threads():
                B_red = T2.T @ np.matmul(B[:, 0], Tn).reshape(nU * 6, nDOF)    # sum_u T_u^T B_u T_u
            F_red = _real_times_complex(T2.T, np.moveaxis(F[:, 0], 1, 0).reshape(nH, nU * 6, nw))
        fowt.B_hydro_drag = B_red
        fowt._raftx_Fdrag = F_red
        fowt.F_hydro_drag = F_red[0].copy()
        return fowt.B_hydro_drag

    def _solve_general(self, model, case, tol, display):
        """raft_model.py:966-1302 for ONE unit with more than 6 reduced DOFs: the same fixed point with the strip
        sweeps node by node, the projections with T, the nDOF x nDOF impedance solves of every bin and the convergence
        test all on the device (raftx_flex_solve, raft_amd/csrc/raftx_flex.h); the host reduces the inertial excitation
        once.  A unit WITHOUT wet strips (a dry structure) has no drag to linearise: its response is one batch of
        dense solves (raftx_solve_dense), which is also what the reference's loop reduces to (B_hydro_drag = 0)."""
        fowts = model.fowtList
        if len(fowts) != 1 or getattr(model, "ms", None):
            raise UnsupportedFOWT("arrays of units with more than 6 reduced DOFs are not on the device path")
        fowt = fowts[0]
        self._check_supported(fowt)
        nw, n = model.nw, int(fowt.nDOF)
        ctx = self.ctx
        self._excitation_general(fowt, case, fowt.memberList)                # :1002
        if fowt.nrotors > 0:                                                 # :1005-1010
            M_turb = _sum_rotors(fowt.A_aero)
            B_turb = _sum_rotors(fowt.B_aero)
        else:
            M_turb = B_turb = np.zeros([n, n, nw])
        fowt.Fhydro_2nd = np.zeros([fowt.nWaves, n, nw], dtype=complex)      # :1035-1036
        fowt.Fhydro_2nd_mean = np.zeros([fowt.nWaves, n])
        B_gyro = np.sum(fowt.B_gyro, axis=2)
        C_moor, dyn = fowt.C_moor, _dynamic_mooring(fowt)
        M_lin = fowt.M_struc + fowt.A_hydro_morison                          # :1045-1047
        if dyn:                                                              # :1022-1030: M, A, C of the lines about XiStart
            XiLast0 = np.zeros([n, nw], dtype=complex) + model.XiStart
            fowt.updateMooringDynamicMatrices(XiLast0[:6], fowt.S[0, :])
            M6, A6, _, C6 = fowt.ms.getCoupledDynamicMatrices(lines_only=True)
            arm = _mooring_arm(fowt)
            M_moor, C_moor = np.zeros([n, n]), np.zeros([n, n])
            M_moor[:6, :6] = translate_matrix_6to6(M6, arm) + translate_matrix_6to6(A6, arm)
            C_moor[:6, :6] = translate_matrix_6to6(C6, arm)
            M_lin = M_lin + M_moor
        B_lin = fowt.B_struc + B_gyro
        C_lin = fowt.C_struc + fowt.C_hydro + C_moor + fowt.C_elast
        A_BEM, B_BEM = np.asarray(getattr(fowt, "A_BEM", 0.0)), np.asarray(getattr(fowt, "B_BEM", 0.0))
        if _nonzero(M_turb) or _nonzero(A_BEM):
            M_lin = M_turb + M_lin[:, :, None] + (A_BEM if _nonzero(A_BEM) else 0.0)       # :1045
        if _nonzero(B_turb) or _nonzero(B_BEM):
            B_lin = B_turb + B_lin[:, :, None] + (B_BEM if _nonzero(B_BEM) else 0.0)       # :1046
        F_lin = fowt.F_BEM[0] + fowt.F_hydro_iner[0] + fowt.Fhydro_2nd[0]    # :1048
        # the fixed point itself runs on the device (raftx_flex_solve): node motions, strip linearisation of every node,
        # the projections with T, the dense solves and the convergence test; M, C and the iterate-independent part of B
        # (with the rotor's frequency-dependent matrices [n,n,nw]: 7.2 MB each at 150 DOFs x 40 bins) go up once
        rows, tables, Tn = fowt._raftx_nodes
        nH = fowt.nWaves
        F_lin = fowt.F_BEM + fowt.F_hydro_iner + fowt.Fhydro_2nd             # :1048, 1212 without the drag excitation
        if dyn and not tables:
            raise UnsupportedFOWT("moorMod==2 on a unit with %d reduced DOFs and no wet strips" % n)
        if tables and dyn:
            # :1069-1072: the lines' damping about every iterate -- a host step between iterations, so the fixed point is
            # stepped: one launch per iteration (loop bound 1) from an explicit linearisation point (raftx_flex_start), the
            # relaxation of :1133 here; the launch of the last iteration leaves every heading's response and Z as :1155-1236 do
            self._upload([fowt], fowt.zeta, fowt.beta, tables=tables)
            XiLast = np.zeros([1, 1, n, nw], dtype=complex) + model.XiStart
            conv, nit, out = False, 0, None
            for iiter in range(int(model.nIter) + 1):                         # :977
                fowt.updateMooringDynamicMatrices(XiLast[0, 0, :6, :], fowt.S[0, :])          # :1070
                _, _, B6, _ = fowt.ms.getCoupledDynamicMatrices(lines_only=True)
                B_moor = np.zeros([n, n])
                B_moor[:6, :6] = translate_matrix_6to6(B6, arm)                              # :1072
                B_it = B_lin + (B_moor if B_lin.ndim == 2 else B_moor[:, :, None])            # :1079
                ctx.flex_start(XiLast)
                out = ctx.flex_solve([0, len(rows)], Tn, M_lin[None], B_it[None], C_lin[None], F_lin[None, None], 0, tol,
                                     model.XiStart, want_Z=True)
                nit = iiter + 1
                if int(out["flags"][0, 0]) & 2:
                    break
                if int(out["flags"][0, 0]) & 1:
                    conv = True
                    break
                XiLast[0, 0] = 0.2 * XiLast[0, 0] + 0.8 * out["Xi"][0, 0, 0]                 # :1133
            out["niter"][0, 0] = nit
            out["flags"][0, 0] = (int(out["flags"][0, 0]) & ~1) | (1 if conv else 0)
        elif tables:
            self._upload([fowt], fowt.zeta, fowt.beta, tables=tables)         # no-op while these tables are resident
            out = ctx.flex_solve([0, len(rows)], Tn, M_lin[None], B_lin[None], C_lin[None], F_lin[None, None], int(model.nIter), tol,
                                 model.XiStart, want_Z=True)
        else:
            # no wet strips: B_hydro_drag = 0 and F_hydro_drag = 0 in every iteration (raft_fowt.py:1905-1936 sums nothing),
            # so the loop of :1058-1138 solves the same systems until two successive responses agree -- the second pass
            Xi_d, Z_d = ctx.solve_dense(model.w, M_lin, B_lin, C_lin, F_lin, want_Z=True)
            # XiLast_k = Xi + 0.2^(k-1) (XiStart - Xi) (:1133 with a constant Xi): the count of :1052,1103 follows in closed form
            gap = np.abs(Xi_d[0] - model.XiStart) / (np.abs(Xi_d[0]) + tol)
            nit, ok = 0, False
            while nit < int(model.nIter) + 1 and not ok:                    # nIter + 1 passes at most (raft_model.py:977)
                ok = bool(np.all(gap * 0.2 ** nit < tol))
                nit += 1
            first_ok = ok
            bad = not np.all(np.isfinite(Xi_d.view(float)))
            out = {"niter": np.array([[nit]], dtype=np.int32),
                   "flags": np.array([[2 if bad else (1 if first_ok else 0)]], dtype=np.int32),
                   "B_drag": np.zeros([1, 1, n, n]), "F_drag": np.zeros([1, 1, nH, n, nw], dtype=complex),
                   "Xi": Xi_d[None, None], "Z": Z_d[None, None]}
        niter = int(out["niter"][0, 0])
        if int(out["flags"][0, 0]) & 2:
            raise Exception("Nan detected in response vector Xi.")           # :1098-1099
        converged = bool(int(out["flags"][0, 0]) & 1)
        if converged and display > 1:
            print(f" Iteration {niter - 1}, converged (within {tol})")
        if display > 0 and not converged:
            print("WARNING - solveDynamics iteration did not converge to the tolerance.")
        fowt.B_hydro_drag = out["B_drag"][0, 0]
        fowt._raftx_Fdrag = out["F_drag"][0, 0]
        model.Xi = np.zeros([nH + 1, model.nDOF, nw], dtype=complex)         # :1195
        model.Xi[:nH], fowt.Z = out["Xi"][0, 0], out["Z"][0, 0]              # Z: the impedance of the last iteration (:1155)
        fowt.F_hydro_drag = fowt._raftx_Fdrag[nH - 1].copy()
        fowt.Xi = model.Xi[:, :n, :]                                         # :1251-1255
        fowt.Xi_fullDOF = _real_times_complex(fowt.T, fowt.Xi)
        model.results['response'] = {}                                       # :1300
        model._raftx_niter = np.array([niter], dtype=np.int32)
        model._raftx_flags = np.array([1 if converged else 0], dtype=np.int32)
        self._resident = None
        self._general_solved = fowt                                          # saveTurbineOutputs of this unit: from fowt.Xi (host)
        return model.Xi

    def _bem_excitation(self, fowt):
        """Potential-flow excitation of the sea state just set on ``fowt`` (raft_fowt.py:1788-1849,1887): the blend of the
        two neighbouring BEM headings, the rotation back out of the wave-heading frame, the wave amplitudes and the
        array phase run on the device (raftx_bem_excitation) for the unit + sea state currently uploaded; the host only
        applies the unit's rigid reduction T.  Returns (F_BEM [nWaves,nDOF,nw], F_BEM_fullDOF [nWaves,nFullDOF,nw])."""
        self._bem_excitation_units([fowt])
        return fowt.F_BEM, fowt.F_BEM_fullDOF

    def _bem_excitation_units(self, fowts, n_pad=0):
        """F_BEM / F_BEM_fullDOF of the units resident on the context (Model.solveDynamics): one device launch for all
        units that carry potential-flow coefficients and share a BEM heading grid (as the units of a farm do), one more
        per further grid."""
        pot = [bool(getattr(f, "potMod", False) or getattr(f, "potModMaster", 1) in [2, 3]) for f in fowts]
        F6 = None
        if any(pot):
            # one launch per DISTINCT heading grid (the units of a farm usually share one): every unit is interpolated
            # between the neighbours of its own grid, exactly as raft_fowt.py:1804-1831 does per FOWT
            nw = fowts[0].nw
            groups = []                                                     # [(grid, [unit indices])]
            for i, (f, p) in enumerate(zip(fowts, pot)):
                if not p:
                    continue
                hgrid = np.asarray(f.BEM_headings, dtype=float)
                for g_, idx in groups:
                    if g_.shape == hgrid.shape and np.array_equal(g_, hgrid):
                        idx.append(i)
                        break
                else:
                    groups.append((hgrid, [i]))
            adj = [float(getattr(f, "heading_adjust", 0.0)) for f in fowts] + [0.0] * n_pad
            xy = [[float(f.x_ref), float(f.y_ref)] for f in fowts] + [[0.0, 0.0]] * n_pad
            for hgrid, idx in groups:
                X = np.zeros((len(fowts) + n_pad, len(hgrid), 6, nw), dtype=complex)   # n_pad: further designs resident beside the units
                for i in idx:
                    X[i] = np.asarray(fowts[i].X_BEM)[:, :6, :]
                Fg = self.ctx.bem_excitation(hgrid, X, heading_adjust=adj, xy_ref=xy, fetch=True)
                if F6 is None:
                    F6 = np.zeros_like(Fg)
                F6[idx] = Fg[idx]
        for i, f in enumerate(fowts):
            nFull = int(getattr(f, "nFullDOF", f.nDOF))
            F_full = np.zeros([f.nWaves, nFull, f.nw], dtype=complex)
            if F6 is not None and pot[i]:
                F_full[:, :6, :] = F6[i, 0]
            if F6 is None or not pot[i]:                                    # strip-theory unit: zeros, without the reduction
                f.F_BEM = np.zeros([f.nWaves, int(f.nDOF), f.nw], dtype=complex)
            else:
                T = np.asarray(f.T, dtype=float)
                f.F_BEM = np.einsum("fd,hfw->hdw", T, F_full) if T.shape[0] == nFull else F_full[:, :f.nDOF, :].copy()
            f.F_BEM_fullDOF = F_full

    def _upload(self, fowts, case_zeta, case_beta, mats=None, tables=None):
        """Upload N units + one sea state.  mats: per-unit (M0,B0,C0,MBw).  Skipped when exactly these tables, matrices
        and sea state are the ones resident on the context (repeated calcHydroLinearization calls of one fixed point)."""
        f0 = fowts[0]
        nw = len(f0.w)
        tables = [f._raftx_table for f in fowts] if tables is None else tables
        nD = len(tables)
        zeta, beta = np.asarray(case_zeta, dtype=float), np.asarray(case_beta, dtype=float)
        key = getattr(self, "_up_key", None)
        # skip only if nothing else has been uploaded to the (possibly shared) context since: its generation counter
        # moves with every upload / build / crossing, whoever made it
        if (mats is None and key is not None and key[0] is self.ctx and len(key[1]) == nD
                and all(a is b for a, b in zip(key[1], tables)) and key[2].shape == zeta.shape
                and np.array_equal(key[2], zeta) and np.array_equal(key[3], beta)
                and key[4] == self.ctx.resident_generation):
            return
        self._up_key = None
        if mats is None:
            M0 = B0 = C0 = np.zeros((nD, 6, 6))
            MBw = None
        else:
            M0 = np.array([m[0] for m in mats])
            B0 = np.array([m[1] for m in mats])
            C0 = np.array([m[2] for m in mats])
            if any(m[3] is not None for m in mats):
                MBw = np.array([m[3] if m[3] is not None else np.zeros((2, 6, 6, nw)) for m in mats])
            else:
                MBw = None
        ctx = self.ctx
        ctx.upload_designs(tables, M0, B0, C0, nw, MBw)
        # pDyn uses Member.computeWaveKinematics' own defaults rho=1025, g=9.81
        # (raft_member.py:1899; raft_fowt.py:1857 does not forward rho/g)
        ctx.upload_cases(f0.w, f0.k, f0.depth, 1025.0, 9.81, zeta[None, :, :], beta[None, :])
        if mats is None:
            self._up_key = (ctx, list(tables), zeta.copy(), beta.copy(), ctx.resident_generation)

    # ------------------------------------------------------------------ submerged rotors (raft_fowt.py:1780-1784,1861-1883)
    @staticmethod
    def _rotor_kinematics(fowt):
        """rot.u / ud / pDyn of every rotor: zeros, and for a submerged rotor the wave kinematics at its hub for every
        heading (host: one point)."""
        for rot in getattr(fowt, "rotorList", []):
            rot.u = np.zeros([fowt.nWaves, 3, fowt.nw], dtype=complex)
            rot.ud = np.zeros([fowt.nWaves, 3, fowt.nw], dtype=complex)
            rot.pDyn = np.zeros([fowt.nWaves, fowt.nw], dtype=complex)
            if rot.r3[2] < 0:
                for ih in range(fowt.nWaves):
                    rot.u[ih], rot.ud[ih], rot.pDyn[ih] = waves.wave_kin(fowt.zeta[ih], fowt.beta[ih], fowt.w, fowt.k, fowt.depth, rot.r3)

    @staticmethod
    def _rotor_tables(fowt, full_dof=True):
        """Pseudo-strip tables of the unit's submerged rotors (raft_amd/strips.py pack_rotors), or []: first ONE table about
        the reduced-DOF point with every submerged rotor in it (their forces add in the reduced vector); with
        ``full_dof`` then one table about the PRP PER submerged rotor (each rotor's vector goes into its own node's slots of
        the full-DOF array, raft_fowt.py:1864-1883)."""
        from .strips import pack_rotors
        t_red = pack_rotors(fowt, with_node_arm=True)
        if t_red is None:
            return []
        out = [t_red]
        if full_dof:
            out += [pack_rotors(fowt, with_node_arm=False, only=ir) for ir, rot in enumerate(fowt.rotorList) if rot.r3[2] < 0]
        return out

    @staticmethod
    def _add_rotor_excitation(fowt, F_tabs, F_iner, F_full=None):
        """Upstream adds the rotor force to ONE heading only -- the force loop sits behind the heading loop and uses its
        last index (raft_fowt.py:1868-1883): reproduced.  F_tabs: device excitation [nWaves,6,nw] of the tables of
        ``_rotor_tables`` in their order (reduced first, then one per submerged rotor); F_full: the full-DOF array whose
        rotor-node slots take each rotor's PRP-referred vector."""
        ih = fowt.nWaves - 1
        F_iner[ih] += F_tabs[0][ih]
        if F_full is not None:
            wet = [rot for rot in fowt.rotorList if rot.r3[2] < 0]
            for rot, F_prp in zip(wet, F_tabs[1:]):
                node = rot.nodeList[0]
                nd = int(getattr(node, "nDOF", 6))
                i0 = int(getattr(node, "id", 0)) * nd
                if F_full.shape[1] < i0 + 6:              # T^T F_full must stay equal to the reduced vector: never skip silently
                    raise UnsupportedFOWT("submerged rotor on node %d: its slots %d..%d lie outside the %d full DOFs of the unit"
                                          % (int(getattr(node, "id", 0)), i0, i0 + 5, F_full.shape[1]))
                F_full[ih, i0:i0 + 6, :] += F_prp[ih]

    # ------------------------------------------------------------------
    def calcHydroExcitation(self, fowt, case, memberList=[]):
        """raft_fowt.py:1732-1888.  As upstream, only the members of ``memberList`` contribute strip-theory excitation
        (the default empty list gives none: Model.solveDynamics passes fowt.memberList, raft_model.py:1017).  Sets nWaves,
        beta, S, zeta, F_BEM(_fullDOF), F_hydro_iner(_fullDOF)."""
        self._check_supported(fowt)
        if _general(fowt):
            self._no_materialise_general(fowt)
            return self._excitation_general(fowt, case, list(memberList))
        self._sea_state(fowt, case)
        members = list(memberList)
        nw, nFull = fowt.nw, int(getattr(fowt, "nFullDOF", fowt.nDOF))
        fowt._raftx_table = pack_fowt(fowt, members)
        # per-member tables about each member's own node give the full-DOF vector (one extra design per member in the
        # same launch); reference objects number their nodes, stand-ins without node ids get the reduced vector only
        ids = [getattr(m.nodeList[0], "id", None) for m in members]
        per_member = []
        if members and all(i is not None for i in ids):
            per_member = [pack_fowt(fowt, [m], own_node=True) for m in members]
        self._rotor_kinematics(fowt)
        rotor_tables = self._rotor_tables(fowt)
        self._upload([fowt], fowt.zeta, fowt.beta, tables=[fowt._raftx_table] + per_member + rotor_tables)
        self._up_key = None                                  # more designs than the unit's own table are resident
        if members or rotor_tables:
            F = self.ctx.excitation()[:, 0]                  # [1 + nMembers (+ 2), nWaves, 6, nw]
        else:
            F = np.zeros([1, fowt.nWaves, 6, nw], dtype=complex)
        if self.materialise_members and members:             # what Member.calcHydroExcitation leaves on each member
            self._materialise_kinematics(fowt, members, design=0)
            fowt._raftx_members = members
            for i, m in enumerate(members):
                if per_member:
                    m.F_hydro_iner = np.ascontiguousarray(F[1 + i])
        fowt.F_hydro_iner = np.ascontiguousarray(F[0])
        fowt.F_hydro_iner_fullDOF = np.zeros([fowt.nWaves, nFull, nw], dtype=complex)
        if per_member:
            for i, m in enumerate(members):
                node = m.nodeList[0]
                i0 = int(node.id) * int(getattr(node, "nDOF", 6))
                fowt.F_hydro_iner_fullDOF[:, i0:i0 + 6, :] += F[1 + i]
        elif nFull == 6:
            fowt.F_hydro_iner_fullDOF[:] = F[0]
        if rotor_tables:                                     # :1861-1883 (device sweep over the rotor's pseudo-strips)
            n0 = 1 + len(per_member)
            self._add_rotor_excitation(fowt, [F[n0 + j] for j in range(len(rotor_tables))], fowt.F_hydro_iner, fowt.F_hydro_iner_fullDOF)
        # potential-flow part: needs the unit's own table + sea state resident (one design)
        self._upload([fowt], fowt.zeta, fowt.beta)
        fowt.F_BEM, fowt.F_BEM_fullDOF = self._bem_excitation(fowt)
        fowt._raftx_fresh = True
        return None

    def calcHydroLinearization(self, fowt, Xi):
        """raft_fowt.py:1891-1936 (heading 0 only, :1910)."""
        if _general(fowt):
            if not hasattr(fowt, "_raftx_nodes"):
                raise RuntimeError("calcHydroExcitation must be called before calcHydroLinearization")
            return self._linearization_general(fowt, Xi)
        if not hasattr(fowt, "_raftx_table"):
            raise RuntimeError("calcHydroExcitation must be called before calcHydroLinearization")
        self._upload([fowt], fowt.zeta, fowt.beta)          # no-op while this unit and sea state are resident
        B, F = self.ctx.linearize(np.asarray(Xi, dtype=complex)[None, None, :, :])
        fowt.B_hydro_drag = B[0, 0]
        fowt._raftx_Fdrag = F[0, 0]                         # [nWaves,6,nw]
        fowt.F_hydro_drag = F[0, 0, 0].copy()
        if self.materialise_members and getattr(fowt, "_raftx_members", None):
            # mem.Bmat of this linearisation and mem.F_exc_drag of heading 0 (raft_fowt.py:1910; raft_member.py:2117,2122)
            self._materialise_drag(fowt, fowt._raftx_members, np.asarray(Xi, dtype=complex)[:6], 0)
        return fowt.B_hydro_drag

    def calcDragExcitation(self, fowt, ih):
        """raft_fowt.py:1940-1957."""
        if not hasattr(fowt, "_raftx_Fdrag"):
            raise RuntimeError("calcHydroLinearization must be called before calcDragExcitation")
        fowt.F_hydro_drag = fowt._raftx_Fdrag[ih].copy()
        if self.materialise_members and getattr(fowt, "_raftx_members", None) and hasattr(fowt._raftx_members[0], "Bmat"):
            for mem in fowt._raftx_members:                  # raft_member.py:2146: F_exc_drag <- Bmat u[ih]
                mem.F_exc_drag = np.einsum("sab,sbw->saw", mem.Bmat, mem.u[ih])
        return fowt.F_hydro_drag

    # ------------------------------------------------------------------ second-order loads
    def calcQTF_slenderBody(self, fowt, waveHeadInd, Xi0=None, verbose=False, iCase=None, iWT=None):
        """raft_fowt.py:1988-2078: fowt.qtf [nw2,nw2,1,nDOF] for heading fowt.beta[waveHeadInd]."""
        import os
        from . import qtf as rq
        w2, k2 = np.asarray(fowt.w1_2nd, dtype=float), np.asarray(fowt.k1_2nd, dtype=float)
        if Xi0 is None:
            Xi0 = np.zeros([fowt.nDOF, len(fowt.w)], dtype=complex)
        beta = float(fowt.beta[waveHeadInd])
        fowt.heads_2nd = [beta]
        fowt.qtf = np.zeros([len(w2), len(w2), 1, fowt.nDOF], dtype=complex)
        if fowt.nDOF > 6:
            print("Function calcQTF_slenderBody() is not implemented for flexible/multibody FOWTs yet. "
                  "Considering null qtf matrices for now.")
            return
        Xi = np.zeros([6, len(w2)], dtype=complex)
        for iDoF in range(6):
            Xi[iDoF, :] = np.interp(w2, fowt.w, Xi0[iDoF, :], left=0, right=0)          # :2022-2024
        whead = f"{np.degrees(beta) % 360:.2f}".replace('.', 'p')
        out_dir = getattr(fowt, "outFolderQTF", None)
        tag = f"_Case{iCase + 1}_WT{iWT}" if isinstance(iCase, int) and isinstance(iWT, int) else ""
        if out_dir is not None and verbose:
            rq.write_rao4(os.path.join(out_dir, f"raos-slender_body_Head{whead}{tag}.4"), w2, beta, Xi)
        tab = rq.pack_qtf(fowt)
        args = ([tab], Xi[None], np.array([beta]), w2, k2, fowt.depth, fowt.rho_water, fowt.g,
                np.asarray(fowt.M_struc, dtype=float)[None])
        if self._qtf_backend is None and self.ctx.rlib.is_device:
            # Kim & Yue table of the MacCamy-Fuchs members on the device too (raftx_qtf_kay; 1.4 s of SciPy Hankel
            # sums per 200 x 200 grid on the host), consumed by the QTF launch that follows
            if tab.kay_geom:
                self.ctx.qtf_kay([tab], np.array([beta]), w2, k2, fowt.depth, fowt.rho_water, fowt.g)
            q = self.ctx.qtf_slender(*args, None)[0]
        else:
            kay = rq.kay_correction(tab.kay_geom, w2, k2, beta, fowt.depth, rho=fowt.rho_water, g=fowt.g)
            backend_fn = self._qtf_backend or (lambda *a: self.ctx.qtf_slender(*a))
            q = backend_fn(*args, kay[None])[0]
        # the matrix has ONE heading slot (:2014-2015 allocates [nw2, nw2, 1, nDOF]); upstream indexes it with waveHeadInd
        # and therefore raises IndexError for waveHeadInd > 0 -- here the single slot holds the heading asked for
        fowt.qtf[:, :, 0, :] = q
        if out_dir is not None and verbose:
            rq.write_qtf12d(os.path.join(out_dir, f"qtf-slender_body-total_Head{whead}{tag}.12d"), fowt.qtf, w2,
                            fowt.heads_2nd, fowt.rho_water, fowt.g)

    def calcHydroForce_2ndOrd(self, fowt, beta, S0, iCase=None, iWT=None, interpMode='qtf'):
        """raft_fowt.py:2158-2253 (host: interpolation + reductions over the QTF diagonals)."""
        import os
        from . import qtf as rq
        heads = list(fowt.heads_2nd)
        if beta < heads[0]:
            print(f"Warning in calcHydroForce_2ndOrd: angle {beta} is less than the minimum incidence angle in the QTF. "
                  f"An incidence of {heads[0]} will be considered for 2nd order loads.")
        if beta > heads[-1]:
            print(f"Warning in calcHydroForce_2ndOrd: angle {beta} is more than the maximum incidence angle in the QTF. "
                  f"An incidence of {heads[-1]} will be considered for 2nd order loads.")
        q = rq.interp_heading(fowt.qtf, heads, beta)
        if interpMode == 'spectrum':
            f_mean, f = rq.hydro_force_2nd_spectrum(q, np.asarray(fowt.w1_2nd), fowt.w, fowt.dw, S0)
        elif self._qtf_backend is None:                     # device: bilinear interpolation + diagonal sums (raftx_qtf_force)
            fm, ff = self.ctx.qtf_force(np.asarray(fowt.w1_2nd), fowt.w, fowt.dw, np.asarray(S0, dtype=float)[None],
                                        qtf=np.ascontiguousarray(q)[None])
            f_mean, f = fm[0], ff[0]
        else:
            f_mean, f = rq.hydro_force_2nd(q, np.asarray(fowt.w1_2nd), fowt.w, fowt.dw, S0)
        out_dir = getattr(fowt, "outFolderQTF", None)
        if out_dir is not None:
            with open(os.path.join(out_dir, f'f_2nd-_Case{ iCase+1 }_WT{ iWT }.txt'), 'w') as file:
                for w, frow in zip(fowt.w, f.T):
                    file.write(f'{w:.5f} {frow[0]:.5f} {frow[1]:.5f} {frow[2]:.5f} {frow[3]:.5f} {frow[4]:.5f} {frow[5]:.5f}\n')
        return f_mean, f

    # ------------------------------------------------------------------
    def saveTurbineOutputs(self, fowt, results, case):
        """raft_fowt.py:2291-2745 for a rigid single unit whose responses are resident on the ctx (i.e. right after
        solveDynamics): platform motions, nacelle accelerations and the tower-base bending moment -- every
        getRMS / getPSD of the method -- are ONE statistics launch over the resident responses
        (raftx_channel_stats_poly); means, +-3 sigma bounds and the response amplitudes are host scalars / views.
        Quasi-static mooring tensions (moorMod == 0, :2356-2399) ride on the same launch: the tension Jacobian and the mean
        tensions are asked of the unit's MoorPy system on the host (third-party input, a dense [2 nLines, 6] matrix), the
        amplitudes J Xi and their statistics are channels of the device.  Line-dynamics tensions (moorMod != 0: MoorPy's
        own frequency-domain line solver per line) and rotor-controller blocks (CCBlade state) stay outside."""
        if _general(fowt):
            if getattr(self, "_general_solved", None) is not fowt:
                raise UnsupportedFOWT("saveTurbineOutputs: this engine has not solved this FOWT (call solveDynamics of its "
                                      "single-unit model first)")
            return self._save_outputs_general(fowt, results, case)
        if getattr(self, "_resident", None) is not fowt:
            raise UnsupportedFOWT("saveTurbineOutputs: the responses of this FOWT are not resident on the device "
                                  "(call solveDynamics of its single-unit model first)")
        J_moor, T_moor = _quasi_static_tension_rows(fowt)
        if any(getattr(rot, "aeroServoMod", 0) > 1 for rot in fowt.rotorList):
            raise UnsupportedFOWT("saveTurbineOutputs: rotor-controller outputs (raft_fowt.py:2640-2680) are not on the device path")
        if np.any(np.abs(np.asarray(fowt.rigidBodyNode.r0[:3], dtype=float)) > 0):
            raise UnsupportedFOWT("saveTurbineOutputs: reference node away from the PRP")
        nr, nw = int(fowt.nrotors), fowt.nw
        deg = 57.29577951308232                                              # helpers.rad2deg
        Lt, Gt, info = tower_base_rows(fowt)
        nT = 0 if J_moor is None else J_moor.shape[0]
        nCh = 6 + 4 * nr + nT
        L = np.zeros((nCh, 3, 6))
        if nT:
            L[6 + 4 * nr:, 0, :] = J_moor                                    # tension amplitudes J Xi_PRP (:2367)
        for j in range(6):
            L[j, 0, j] = 1.0 if j < 3 else deg                               # motions; rotations in degrees (:2332-2354)
        for ir, rotor in enumerate(fowt.rotorList):
            T = np.asarray(rotor.nodeList[0].T, dtype=float)                 # hub motion = T Xi (:2423)
            L[6 + 3 * ir:9 + 3 * ir, 2, :] = T[:3, :]                        # accelerations: w^2 x (:2426-2441)
            L[6 + 3 * nr + ir] = Lt[ir]
        Gw = None
        if Gt is not None:
            Gw = np.zeros((nCh, 6, nw), dtype=complex)
            Gw[6 + 3 * nr:6 + 4 * nr] = Gt
        std, psd = self.ctx.channel_stats_poly(L, fowt.dw, Gw=Gw, want_psd=True)
        std, psd = std[0, 0], psd[0, 0]
        Xi0 = np.asarray(fowt.r6, dtype=float) - np.array([fowt.x_ref, fowt.y_ref, 0, 0, 0, 0])
        Xi = np.asarray(fowt.Xi)
        for j, name in enumerate(("surge", "sway", "heave", "roll", "pitch", "yaw")):
            avg = Xi0[j] if j < 3 else Xi0[j] * deg
            results[name + "_avg"] = avg
            results[name + "_std"] = std[j]
            results[name + "_max"] = avg + 3 * std[j]
            results[name + "_min"] = avg - 3 * std[j]
            results[name + "_PSD"] = psd[j].copy()
            results[name + "_RA"] = Xi[:, j, :] if j < 3 else Xi[:, j, :] * deg
        for ax_i, ax in enumerate("xyz"):
            key = "A%sRNA" % ax
            for suffix in ("std", "avg", "max", "min"):
                results["%s_%s" % (key, suffix)] = np.zeros(nr)
            results[key + "_PSD"] = np.zeros([nw, nr])
            for ir, rotor in enumerate(fowt.rotorList):
                c = 6 + 3 * ir + ax_i
                rn = rotor.nodeList[0].r
                avg = abs(np.sin(rn[4]) * fowt.g) if ax == "x" else (abs(np.sin(rn[3]) * fowt.g) if ax == "y" else abs(fowt.g))
                results[key + "_std"][ir] = std[c]
                results[key + "_PSD"][:, ir] = psd[c]
                results[key + "_avg"][ir] = avg
                results[key + "_max"][ir] = avg + 3 * std[c]
                results[key + "_min"][ir] = avg - 3 * std[c]
        for base in ("Mbase", "FbaseX", "FbaseY", "FbaseZ", "MbaseX", "MbaseY", "MbaseZ"):
            for suffix in ("avg", "std", "max", "min"):
                results["%s_%s" % (base, suffix)] = np.zeros(nr)
            results[base + "_PSD"] = np.zeros([nw, nr])
        for ir, rotor in enumerate(fowt.rotorList):
            c = 6 + 3 * nr + ir
            m, hArm = info[ir]
            f = np.asarray(fowt.rotorList[0].nodeList[0].T, dtype=float) @ np.asarray(fowt.f_aero0)[:, ir]
            results["Mbase_avg"][ir] = m * fowt.g * hArm * np.sin(fowt.Xi0[4]) + (f[4] + (-hArm) * f[0])     # :2532-2533
            results["Mbase_std"][ir] = std[c]
            results["Mbase_PSD"][:, ir] = psd[c]
            results["Mbase_max"][ir] = results["Mbase_avg"][ir] + 3 * std[c]
            results["Mbase_min"][ir] = results["Mbase_avg"][ir] - 3 * std[c]
        if nT:
            c0 = 6 + 4 * nr
            self._put_tensions(results, T_moor, std[c0:c0 + nT], psd[c0:c0 + nT], fowt)
        zeta = np.asarray(fowt.zeta)
        results["wave_PSD"] = np.sum(0.5 * np.abs(zeta) ** 2 / fowt.dw, axis=0)                       # getPSD(zeta, dw)
        for key in ("omega", "torque", "bPitch"):
            results[key + "_avg"] = np.zeros(nr)
            results[key + "_std"] = np.zeros(nr)
            results[key + "_PSD"] = np.zeros([nw, nr])
        for key in ("omega_max", "omega_min", "power_avg"):
            results[key] = np.zeros(nr)
        return results

    @staticmethod
    def _put_tensions(results, T_moor, std_t, psd_t, fowt):
        results["Tmoor_avg"] = T_moor
        results["Tmoor_std"] = np.array(std_t)
        results["Tmoor_max"] = T_moor + 3 * np.asarray(std_t)
        results["Tmoor_min"] = T_moor - 3 * np.asarray(std_t)
        results["Tmoor_PSD"] = np.asarray(psd_t) * (fowt.dw / fowt.w[0])    # (sic) getPSD(.., self.w[0]), raft_fowt.py:2372,2399

    def _save_outputs_general(self, fowt, results, case):
        """raft_fowt.py:2291-2745 for a single unit with MORE than six reduced DOFs (flexible members): every getRMS /
        getPSD of the method is a linear channel of the reduced response through the rows of T -- PRP motions from the
        rigid-body node (:2299-2355), hub accelerations (:2422-2444) and the tower-base loads: the finite-element
        internal loads -Kf Xi_internal at the base node of a FLEXIBLE tower (:2540-2601), the fore-aft moment formula
        (:2500-2537) of a rigid one -- ONE statistics launch over the response the solve returned
        (raftx_response_stats).  Means and +-3 sigma bounds are host scalars.  Quasi-static mooring tensions (:2356-2399) are
        rows J_moor over the PRP motion rows."""
        J_moor, T_moor = _quasi_static_tension_rows(fowt)
        if any(getattr(rot, "aeroServoMod", 0) > 1 for rot in fowt.rotorList):
            raise UnsupportedFOWT("saveTurbineOutputs: rotor-controller outputs (raft_fowt.py:2640-2680) are not on the device path")
        missing = [a for a in ("rigidBodyNode", "memberList", "rotorList", "T", "r6", "nplatmems") if not hasattr(fowt, a)]
        if missing:
            raise UnsupportedFOWT("saveTurbineOutputs: this FOWT object does not carry %s (a stand-in without the node / member "
                                  "structure the output channels are built from)" % ", ".join(missing))
        nr, nw, n = int(fowt.nrotors), fowt.nw, int(fowt.nDOF)
        deg = 57.29577951308232                                              # helpers.rad2deg
        T = np.asarray(fowt.T, dtype=float)                                  # [nFullDOF, nDOF]
        nFull = T.shape[0]
        Xi = np.ascontiguousarray(fowt.Xi, dtype=complex)                    # [nWaves + 1, nDOF, nw]
        w = np.asarray(fowt.w, dtype=float)
        rows_full, rows_red, gw_red = [], [], {}                             # channel -> (power, row over full DOFs) / reduced

        def full_row(p, cols, vals):
            r = np.zeros((3, nFull))
            r[p, cols] = vals
            rows_full.append(r)
            rows_red.append(None)
            return len(rows_full) - 1
        # platform motions at the PRP from the rigid-body node (:2299-2307): Xi_t + th x (-r0), th
        i0 = int(fowt.rigidBodyNode.id)                                      # (upstream slices [id : id + 6])
        r0 = -np.asarray(fowt.rigidBodyNode.r0[:3], dtype=float)
        A = np.array([[0.0, r0[2], -r0[1]], [-r0[2], 0.0, r0[0]], [r0[1], -r0[0], 0.0]])   # th x r = A th (helpers.py:396-402)
        ch_motion = []
        for j in range(3):
            ch_motion.append(full_row(0, [i0 + j, i0 + 3, i0 + 4, i0 + 5], [1.0, A[j, 0], A[j, 1], A[j, 2]]))
        for j in range(3, 6):
            ch_motion.append(full_row(0, [i0 + j], [deg]))
        ch_tens = []                                                         # tensions J Xi_PRP (:2367), rotations in radians
        if J_moor is not None:
            for Ji in J_moor:
                rot = Ji[:3] @ A + Ji[3:]
                ch_tens.append(full_row(0, [i0, i0 + 1, i0 + 2, i0 + 3, i0 + 4, i0 + 5], [Ji[0], Ji[1], Ji[2], rot[0], rot[1], rot[2]]))
        ch_acc = []                                                          # hub accelerations: w^2 x (:2422-2444)
        for rotor in fowt.rotorList:
            h0 = int(rotor.nodeList[0].id) * 6
            ch_acc.append([full_row(2, [h0 + a], [1.0]) for a in range(3)])
        ch_base, base_mean, rigid_info = [], [], []
        for ir, rotor in enumerate(fowt.rotorList):
            mem_tower = fowt.memberList[fowt.nplatmems + ir]
            if getattr(mem_tower, "type", "rigid") == "rigid":               # :2500-2537 on the reduced DOFs 0 and 4
                Lt, Gt, info = tower_base_rows(fowt, only=ir)
                r = np.zeros((3, n))
                r[:, :6] = Lt[ir]
                rows_full.append(None)
                rows_red.append(r)
                c = len(rows_full) - 1
                if Gt is not None:
                    g = np.zeros((n, nw), dtype=complex)
                    g[:6] = Gt[ir]
                    gw_red[c] = g
                ch_base.append(("rigid", [c]))
                rigid_info.append(info[0])
                base_mean.append(None)
            else:                                                            # :2540-2601: internal loads from the FE stiffness
                Kf = np.asarray(mem_tower.Kf, dtype=float)
                iF, iL = int(mem_tower.nodeList[0].id), int(mem_tower.nodeList[-1].id)
                cols = np.arange(iF * 6, (iL + 1) * 6)
                first = mem_tower.nodeList[0].r0[2] <= mem_tower.nodeList[-1].r0[2]
                base = slice(0, 6) if first else slice(Kf.shape[0] - 6, Kf.shape[0])
                Kb = -Kf[base, :]
                ch_base.append(("flex", [full_row(0, cols, Kb[a]) for a in range(6)]))
                Xi0_int = np.concatenate([np.asarray(nd.Xi0, dtype=float) for nd in mem_tower.nodeList])
                base_mean.append((-Kf @ Xi0_int)[base])
                rigid_info.append(None)
        L = np.array([rr if rr is not None else rf @ T for rf, rr in zip(rows_full, rows_red)])      # [nCh,3,nDOF]
        Gw = None
        if gw_red:
            Gw = np.zeros((len(L), n, nw), dtype=complex)
            for c, g in gw_red.items():
                Gw[c] = g
        std, psd = self.ctx.response_stats(w, L, Xi, fowt.dw, Gw=Gw, want_psd=True)
        Xi0 = np.asarray(fowt.r6, dtype=float) - np.array([fowt.x_ref, fowt.y_ref, 0, 0, 0, 0])
        # response amplitudes of the PRP (the reference stores them): the same rows applied on the host (6 x nDOF x nw)
        Lm = L[ch_motion, 0, :]
        Xi_prp = np.einsum("cd,hdw->hcw", Lm, Xi)
        for j, name in enumerate(("surge", "sway", "heave", "roll", "pitch", "yaw")):
            c = ch_motion[j]
            avg = Xi0[j] if j < 3 else Xi0[j] * deg
            results[name + "_avg"] = avg
            results[name + "_std"] = std[c]
            results[name + "_max"] = avg + 3 * std[c]
            results[name + "_min"] = avg - 3 * std[c]
            results[name + "_PSD"] = psd[c].copy()
            results[name + "_RA"] = Xi_prp[:, j, :]
        if ch_tens:
            self._put_tensions(results, T_moor, std[ch_tens], psd[ch_tens], fowt)
        for ax_i, ax in enumerate("xyz"):
            key = "A%sRNA" % ax
            for suffix in ("std", "avg", "max", "min"):
                results["%s_%s" % (key, suffix)] = np.zeros(nr)
            results[key + "_PSD"] = np.zeros([nw, nr])
            for ir, rotor in enumerate(fowt.rotorList):
                c = ch_acc[ir][ax_i]
                rn = rotor.nodeList[0].r
                avg = abs(np.sin(rn[4]) * fowt.g) if ax == "x" else (abs(np.sin(rn[3]) * fowt.g) if ax == "y" else abs(fowt.g))
                results[key + "_std"][ir] = std[c]
                results[key + "_PSD"][:, ir] = psd[c]
                results[key + "_avg"][ir] = avg
                results[key + "_max"][ir] = avg + 3 * std[c]
                results[key + "_min"][ir] = avg - 3 * std[c]
        bases = ("FbaseX", "FbaseY", "FbaseZ", "MbaseX", "MbaseY", "MbaseZ")
        for base in ("Mbase",) + bases:
            for suffix in ("avg", "std", "max", "min"):
                results["%s_%s" % (base, suffix)] = np.zeros(nr)
            results[base + "_PSD"] = np.zeros([nw, nr])

        def put(base, ir, avg, c):
            results[base + "_avg"][ir] = avg
            results[base + "_std"][ir] = std[c]
            results[base + "_PSD"][:, ir] = psd[c]
            results[base + "_max"][ir] = avg + 3 * std[c]
            results[base + "_min"][ir] = avg - 3 * std[c]
        for ir, rotor in enumerate(fowt.rotorList):
            kind, chans = ch_base[ir]
            if kind == "rigid":
                m, hArm = rigid_info[ir]
                f = np.asarray(fowt.rotorList[0].nodeList[0].T, dtype=float) @ np.asarray(fowt.f_aero0)[:, ir]
                f = transform_force_moment_y(f, hArm)
                put("Mbase", ir, m * fowt.g * hArm * np.sin(fowt.Xi0[4]) + f, chans[0])             # :2532-2533
            else:
                for a, base in enumerate(bases):
                    put(base, ir, base_mean[ir][a], chans[a])
                put("Mbase", ir, base_mean[ir][4], chans[4])                                          # :2594-2599 (= MbaseY)
        zeta = np.asarray(fowt.zeta)
        results["wave_PSD"] = np.sum(0.5 * np.abs(zeta) ** 2 / fowt.dw, axis=0)                       # getPSD(zeta, dw)
        for key in ("omega", "torque", "bPitch"):
            results[key + "_avg"] = np.zeros(nr)
            results[key + "_std"] = np.zeros(nr)
            results[key + "_PSD"] = np.zeros([nw, nr])
        for key in ("omega_max", "omega_min", "power_avg"):
            results[key] = np.zeros(nr)
        return results

    def _solve_stepped(self, model, fowts, mats, F_extra, tol, display, qtf_hook=None):
        """The drag fixed point with a host step between iterations -- raft_model.py:1069-1072: with ``moorMod == 2``
        the mooring system's linearised damping is re-evaluated (by MoorPy, on the host) about every iterate.  One
        device launch per iteration (loop bound 1) from an explicit linearisation point; the relaxation (:1133) is done
        here.  Every unit keeps the reference's OWN loop state (:1052-1142 is a loop per unit): its pass counter, its
        linearisation point, and -- ``qtf_hook(i, Xi_i) -> F_extra_i`` given, for units with internal QTFs
        (potSecOrder == 1) -- the re-entry of :1108-1131: at its first convergence the unit's QTFs and second-order
        force are computed from that response, the force joins its excitation, the pass counter restarts at 1 and the
        loop goes on from the SAME linearisation point.  A unit that has left its loop is frozen: its results are the
        ones of the launch it left with, whatever the slower units of a farm still need."""
        ctx = self.ctx
        nF, nw = len(fowts), model.nw
        f0 = fowts[0]
        nIter = int(model.nIter) + 1                                        # :977
        XiLast = np.zeros([nF, 1, 6, nw], dtype=complex) + model.XiStart    # :999
        B_base = [np.array(m[1], dtype=float) for m in mats]
        F_extra = np.array(F_extra, dtype=complex)
        conv = np.zeros(nF, dtype=bool)
        done = np.zeros(nF, dtype=bool)
        it = np.zeros(nF, dtype=np.int64)                                   # the reference's iiter, per unit
        niter = np.zeros(nF, dtype=np.int32)                                # launches the unit took part in
        qtf_pending = [qtf_hook is not None and getattr(f, "potSecOrder", 0) == 1 for f in fowts]
        final = {}
        out = None
        while not done.all():
            for i, fowt in enumerate(fowts):
                if _dynamic_mooring(fowt) and not done[i]:
                    fowt.updateMooringDynamicMatrices(XiLast[i, 0, :6, :], fowt.S[0, :])        # :1070
                    _, _, B6, _ = fowt.ms.getCoupledDynamicMatrices(lines_only=True)
                    mats[i][1] = B_base[i] + translate_matrix_6to6(B6, _mooring_arm(fowt))        # :1072,:1079
            self._upload(fowts, f0.zeta, f0.beta, mats)
            ctx.set_linearisation_point(XiLast, keep_last=False)
            out = ctx.solve_dynamics(0, tol=tol, XiStart=model.XiStart,
                                     F_extra=F_extra if np.any(F_extra) else None,
                                     want_Xi=True, want_B=True, want_F=True, want_Z=True)
            if np.any(out['flags'][:nF][~done] & 2):
                final = {}                                                  # NaN: raised by the caller (:1098-1099)
                break
            for i in range(nF):
                if done[i]:
                    continue
                niter[i] += 1
                if out['flags'][i, 0] & 1:
                    if qtf_pending[i]:                                      # :1108-1131
                        F_extra[i, 0] = qtf_hook(i, out['Xi'][i, 0, 0])
                        qtf_pending[i] = False
                        it[i] = 0                                           # (:1114, then the common iiter += 1)
                    else:
                        conv[i] = done[i] = True
                else:
                    XiLast[i, 0] = 0.2 * XiLast[i, 0] + 0.8 * out['Xi'][i, 0, 0]               # :1133
                it[i] += 1
                if it[i] >= nIter:
                    done[i] = True
                if done[i]:
                    for key in ('Xi', 'B_drag', 'F_wave', 'Z'):
                        final.setdefault(key, {})[i] = np.array(out[key][i])
        for key, rows in final.items():                                     # every unit: the launch it left its loop with
            for i, v in rows.items():
                out[key][i] = v
        out['niter'] = niter[:, None].copy()
        out['flags'] = (out['flags'] & ~1) | conv[:, None].astype(np.int32)
        out['XiLast'] = XiLast                                              # what the reference's loop holds at its exit (:1156)
        out['F_extra'] = F_extra
        return out

    def solveDynamics(self, model, case, tol=0.01, conv_plot=0, RAO_plot=0, display=0):
        """raft_model.py:966-1302."""
        iCase = case['iCase'] if 'iCase' in case else None
        fowts = model.fowtList
        if any(_general(f) for f in fowts):
            for f in fowts:
                self._no_materialise_general(f)
            return self._solve_general(model, case, tol, display)
        nF = len(fowts)
        nw = model.nw
        mats, F_extras = [], []
        for i, fowt in enumerate(fowts):
            self._check_supported(fowt)
            # sea state + excitation inputs (raft_model.py:1002)
            self._sea_state(fowt, case)
            fp = pack_fingerprint(fowt)                     # analyzeCases: one packing per pose, not per load case
            if getattr(fowt, "_raftx_table_fp", None) != fp or getattr(fowt, "_raftx_table_all", None) is None:
                fowt._raftx_table_all, fowt._raftx_table_fp = pack_fowt(fowt), fp
            fowt._raftx_table = fowt._raftx_table_all

            if fowt.nrotors > 0:                                            # :1005-1010
                M_turb = np.sum(fowt.A_aero, axis=3)
                B_turb = np.sum(fowt.B_aero, axis=3)
            else:
                M_turb = np.zeros([6, 6, nw])
                B_turb = np.zeros([6, 6, nw])
            fowt.Fhydro_2nd = np.zeros([fowt.nWaves, fowt.nDOF, fowt.nw], dtype=complex)   # :1035-1038
            fowt.Fhydro_2nd_mean = np.zeros([fowt.nWaves, fowt.nDOF])
            if getattr(fowt, "potSecOrder", 0) == 2:
                fowt.Fhydro_2nd_mean[0, :], fowt.Fhydro_2nd[0, :, :] = \
                    self.calcHydroForce_2ndOrd(fowt, fowt.beta[0], fowt.S[0, :], iCase=iCase, iWT=i)
                for ih in range(1, fowt.nWaves):                            # :1210-1211
                    fowt.Fhydro_2nd_mean[ih, :], fowt.Fhydro_2nd[ih, :, :] = \
                        self.calcHydroForce_2ndOrd(fowt, fowt.beta[ih], fowt.S[ih, :])
            C_moor = fowt.C_moor
            MA_moor = None
            if _dynamic_mooring(fowt):                                      # :1022-1030
                XiLast0 = np.zeros([fowt.nDOF, nw], dtype=complex) + model.XiStart
                fowt.updateMooringDynamicMatrices(XiLast0[:6], fowt.S[0, :])
                M6, A6, _, C6 = fowt.ms.getCoupledDynamicMatrices(lines_only=True)
                r_moor = _mooring_arm(fowt)
                MA_moor = translate_matrix_6to6(M6, r_moor) + translate_matrix_6to6(A6, r_moor)
                C_moor = translate_matrix_6to6(C6, r_moor)
            A_BEM = np.asarray(fowt.A_BEM)
            B_BEM = np.asarray(fowt.B_BEM)
            B_gyro = np.sum(fowt.B_gyro, axis=2)
            C_lin = fowt.C_struc + fowt.C_hydro + C_moor + fowt.C_elast    # :1047
            if _nonzero(M_turb) or _nonzero(B_turb) or _nonzero(A_BEM) or _nonzero(B_BEM):
                M_lin = M_turb + fowt.M_struc[:, :, None] + A_BEM + fowt.A_hydro_morison[:, :, None]   # :1045
                B_lin = B_turb + fowt.B_struc[:, :, None] + B_BEM + B_gyro[:, :, None]                 # :1046
                mats.append([np.zeros((6, 6)) if MA_moor is None else MA_moor, np.zeros((6, 6)), C_lin,
                             np.array([M_lin, B_lin])])
            else:
                mats.append([fowt.M_struc + fowt.A_hydro_morison + (0.0 if MA_moor is None else MA_moor),
                             fowt.B_struc + B_gyro, C_lin, None])

        f0 = fowts[0]
        # submerged rotors (:1861-1883), any number on any unit of an array: each such unit's pseudo-strip table (all its
        # rotors, about its reduced-DOF point) rides along as one more design (benign matrices), the device evaluates its
        # excitation with everything else and it joins that unit's excitation as part of F_extra
        rotor_tables, rotor_unit = [], []
        for i, f in enumerate(fowts):
            self._rotor_kinematics(f)
            t = self._rotor_tables(f, full_dof=False)
            if t:
                rotor_tables += t
                rotor_unit.append(i)
        if rotor_tables:
            benign = [np.zeros((6, 6)), np.zeros((6, 6)), np.eye(6), None]
            self._upload(fowts, f0.zeta, f0.beta, mats + [benign] * len(rotor_tables), tables=[f._raftx_table for f in fowts] + rotor_tables)
        else:
            self._upload(fowts, f0.zeta, f0.beta, mats)
        ctx = self.ctx
        self._bem_excitation_units(fowts, n_pad=len(rotor_tables))          # F_BEM(_fullDOF) of every unit (:1788-1849,1887)
        F_extras = [fowt.F_BEM + fowt.Fhydro_2nd for fowt in fowts]
        F_iner = ctx.excitation()                                            # side effect of :1002
        F_rotor = {}
        for j, i in enumerate(rotor_unit):
            F_rotor[i] = np.zeros_like(F_iner[0, 0])
            self._add_rotor_excitation(fowts[i], [F_iner[nF + j, 0]], F_rotor[i])
            F_extras[i] = F_extras[i] + F_rotor[i]
        F_extras += [np.zeros_like(F_iner[0, 0])] * len(rotor_tables)
        F_extra = np.array(F_extras)[:, None]                               # [nF (+2),1,nH,6,nw]
        internal_qtf = [getattr(f, "potSecOrder", 0) == 1 for f in fowts]
        if any(internal_qtf) and f0.nWaves > 1:
            # upstream's own branch for further headings is broken (fowt.qtf has a single heading slot, raft_fowt.py:2014,
            # and raft_model.py:1210-1211 would reuse the heading-0 matrix): there is no reference behaviour to match
            raise UnsupportedFOWT("internal slender-body QTFs (potSecOrder == 1) with more than one wave heading are not on "
                                  "the device path")
        array_dynamic = bool(getattr(model, "ms", None)) and getattr(model, "moorMod", 0) == 2
        if any(internal_qtf) or array_dynamic or self.materialise_members:  # the loop's last linearisation point is needed afterwards
            ctx.set_linearisation_point(None, keep_last=True)
        stepped = any(_dynamic_mooring(f) for f in fowts)
        if stepped:
            # moorMod == 2 (:1069-1072), with the re-entry of units that compute their QTFs internally inside the same
            # per-unit loops (:1108-1131)
            def qtf_hook(i, Xi_i):
                fowt = fowts[i]
                if display > 1:
                    print("Resolving for system response in primary wave direction, now with second-order wave loads.")
                Xi0 = waves.get_rao(Xi_i, fowt.zeta[0, :])
                self.calcQTF_slenderBody(fowt, waveHeadInd=0, Xi0=Xi0, verbose=True, iCase=iCase, iWT=i)
                fowt.Fhydro_2nd_mean[0, :], fowt.Fhydro_2nd[0, :, :] = \
                    self.calcHydroForce_2ndOrd(fowt, fowt.beta[0], fowt.S[0, :], iCase=iCase, iWT=i)
                F_extras[i] = fowt.F_BEM + fowt.Fhydro_2nd + F_rotor.get(i, 0.0)
                return F_extras[i]
            out = self._solve_stepped(model, fowts, mats, F_extra, tol, display, qtf_hook=qtf_hook if any(internal_qtf) else None)
        else:
            out = ctx.solve_dynamics(int(model.nIter), tol=tol, XiStart=model.XiStart,
                                     F_extra=F_extra if np.any(F_extra) else None,
                                     want_Xi=True, want_B=True, want_F=True, want_Z=True)
        if any(internal_qtf) and not stepped and not np.any(out['flags'] & 2):
            # raft_model.py:1108-1131: units that converged get their QTFs from the converged first-order motions,
            # the second-order force joins F_lin and the drag iteration continues FROM THE SAME Xi_last with the
            # iteration counter reset to 1.  Units without internal QTFs (or unconverged) simply keep their state:
            # restarting them from their own last linearisation point reproduces their converged solve.
            XiLast = ctx.fetch_linearisation_point()
            rerun = False
            for i, fowt in enumerate(fowts):
                if internal_qtf[i] and (out['flags'][i, 0] & 1):
                    if display > 1:
                        print("Resolving for system response in primary wave direction, now with second-order wave loads.")
                    Xi0 = waves.get_rao(out['Xi'][i, 0, 0], fowt.zeta[0, :])
                    self.calcQTF_slenderBody(fowt, waveHeadInd=0, Xi0=Xi0, verbose=True, iCase=iCase, iWT=i)
                    fowt.Fhydro_2nd_mean[0, :], fowt.Fhydro_2nd[0, :, :] = \
                        self.calcHydroForce_2ndOrd(fowt, fowt.beta[0], fowt.S[0, :], iCase=iCase, iWT=i)
                    for ih in range(1, fowt.nWaves):                                       # :1210-1211
                        fowt.Fhydro_2nd_mean[ih, :], fowt.Fhydro_2nd[ih, :, :] = \
                            self.calcHydroForce_2ndOrd(fowt, fowt.beta[ih], fowt.S[ih, :])
                    F_extras[i] = fowt.F_BEM + fowt.Fhydro_2nd + F_rotor.get(i, 0.0)   # (:1129 adds to F_lin: the rotors' share stays)
                    rerun = True
            if rerun:
                if any(not (internal_qtf[i] and (out['flags'][i, 0] & 1)) for i in range(nF)):
                    raise UnsupportedFOWT("mixed arrays (units with and without converged internal QTFs) are not on the device path")
                F_extra = np.array(F_extras)[:, None]
                ctx.set_linearisation_point(XiLast, keep_last=self.materialise_members)
                niter1 = out['niter'].copy()
                out = ctx.solve_dynamics(max(int(model.nIter) - 1, 0), tol=tol, XiStart=model.XiStart, F_extra=F_extra,
                                         want_Xi=True, want_B=True, want_F=True, want_Z=True)
                out['niter'] = out['niter'] + niter1
        if np.any(out['flags'] & 2):
            raise Exception("Nan detected in response vector Xi.")          # :1098-1099
        nH = f0.nWaves
        for i, fowt in enumerate(fowts):
            fowt.F_hydro_iner = F_iner[i, 0] if i not in F_rotor else F_iner[i, 0] + F_rotor[i]
            fowt.Z = out['Z'][i, 0]                                         # :1155
            fowt.B_hydro_drag = out['B_drag'][i, 0]
            fowt._raftx_Fdrag = out['F_wave'][i, 0] - F_iner[i, 0] - F_extras[i]
            fowt.F_hydro_drag = fowt._raftx_Fdrag[nH - 1].copy()
            if display > 1:
                it = int(out['niter'][i, 0])
                if out['flags'][i, 0] & 1:
                    print(f" Iteration {it - 1}, converged (tolerance {tol})")
            if display > 0 and not (out['flags'][i, 0] & 1):
                print("WARNING - solveDynamics iteration did not converge to the tolerance.")   # :1138-1140

        nDOF = model.nDOF
        model.Xi = np.zeros([nH + 1, nDOF, nw], dtype=complex)              # :1195
        ms = getattr(model, "ms", None)
        if nF == 1 and not ms:
            model.Xi[:nH] = out['Xi'][0, 0]
        else:
            n = 6 * nF
            Mc = Bc = Cc = None
            if ms:                                                          # :1173-1182
                if getattr(model, "moorMod", 0) in (0, 1):
                    Cc = np.asarray(ms.getCoupledStiffnessA(lines_only=True), dtype=float)[None]
                elif model.moorMod == 2:
                    # lumped-mass dynamics of the shared lines, linearised (by MoorPy, on the host) about the motions the
                    # units' loops ended on (:1156,1178): Z_sys += -w^2 (M + A) + i w B + C (:1181-1182) -- the Mc, Bc, Cc
                    # of raftx_solve_system
                    XiLast_all = out['XiLast'] if 'XiLast' in out else ctx.fetch_linearisation_point()
                    # the kernel exports the point the LAST linearisation was made about; a unit that left its loop
                    # unconverged has been relaxed once more upstream (:1133 runs before the loop ends, :1156 appends that)
                    pts = [np.array(XiLast_all[i, 0]) if (out['flags'][i, 0] & 1)
                           else 0.2 * np.asarray(XiLast_all[i, 0]) + 0.8 * np.asarray(out['Xi'][i, 0, 0]) for i in range(nF)]
                    model.updateMooringDynamicMatrices(pts, f0.S[0, :])
                    M_m, A_m, B_m, C_m = (np.asarray(a, dtype=float) for a in ms.getCoupledDynamicMatrices(lines_only=True))
                    Mc, Bc, Cc = (M_m + A_m)[None], B_m[None], C_m[None]
                # any other moorMod: upstream adds zeros (:1174)
            Zblk = out['Z'][:nF, 0][None]                                   # [1,nF,6,6,nw] (rotor pseudo-designs, if any, follow the units)
            Fw = np.transpose(out['F_wave'][:nF, 0], (1, 0, 2, 3)).reshape(1, nH, n, nw)
            model.Xi[:nH] = ctx.solve_system(model.w, Zblk, Fw, Mc=Mc, Bc=Bc, Cc=Cc)[0]
        for i, fowt in enumerate(fowts):                                    # :1251-1255
            fowt.Xi = model.Xi[:, i * fowt.nDOF:(i + 1) * fowt.nDOF, :]
            # T @ Xi[ih] for every heading as ONE real product on the interleaved (re, im) view: `T @ Xi[ih]` with a real T
            # and a complex Xi takes NumPy's mixed-type path (2.3 ms per heading on an 8-core host, 0.75 ms of a 1.36 ms call
            # on the GPU box as an einsum); a real [nFull,6] x [6,2 nw] product is a dgemm
            fowt.Xi_fullDOF = _real_times_complex(fowt.T, fowt.Xi)
        model.results['response'] = {}                                      # :1300
        model._raftx_niter = out['niter'][:, 0].copy()
        if self.materialise_members:
            # what the reference's loop leaves on the members: kinematics of this sea state, Bmat of the LAST linearisation
            # (about the Xi_last the loop exited with, :1063) and F_exc_drag of the last heading's calcDragExcitation (:1214)
            XiL = out['XiLast'] if 'XiLast' in out else ctx.fetch_linearisation_point()
            for i, fowt in enumerate(fowts):
                fowt._raftx_members = list(fowt.memberList)
                self._materialise_kinematics(fowt, fowt._raftx_members, design=i)
                self._materialise_drag(fowt, fowt._raftx_members, np.asarray(XiL[i, 0]), nH - 1, design=i)
        # single-unit models: the unit's responses stay resident on the ctx, which saveTurbineOutputs reads back as
        # statistics; a farm's final responses come from the coupled solve, not from the resident per-unit ones
        self._resident = fowts[0] if nF == 1 else None
        self._general_solved = None
        model._raftx_flags = out['flags'][:, 0].copy()
        return model.Xi


def _general(fowt):
    """More reduced DOFs than the rigid body's six (flexible members, raft_fowt.py's T reduction)."""
    return int(getattr(fowt, "nDOF", 6)) != 6


def _dynamic_mooring(fowt):
    """raft_model.py:1020-1023: the unit has its own mooring system with lumped-mass line dynamics."""
    return bool(getattr(fowt, "ms", None)) and getattr(fowt, "moorMod", 0) == 2


def _mooring_arm(fowt):
    """raft_model.py:1027: from the unit's reduced-DOF reference node to the mooring body's reference point."""
    return np.asarray(fowt.ms.bodyList[0].r6[:3], dtype=float) - np.asarray(fowt.nodeList[fowt.reducedDOF[0][0]].r[:3], dtype=float)


def transform_force_moment_y(f, hArm):
    """[4] of helpers.transformForce(f, offset=[0, 0, -hArm]) (:2533): the fore-aft moment of a 6-vector f about a point
    hArm below its own."""
    f = np.asarray(f, dtype=float)
    return f[4] + (-hArm) * f[0]


def tower_base_rows(fowt, only=None):
    """Tower-base fore-aft bending moment of every (rigid) tower as linear channels of the platform response --
    raft/raft_fowt.py:2500-2528:  M = M_I + M_w + M_X_aero with
        M_w = m g h Xi_pitch,   M_I = -m a_CG h - I_CG (-w^2 Xi_pitch),  a_CG = -w^2 (Xi_surge + z_CG Xi_pitch),
        M_X_aero = -(-w^2 A_aero[0,0] + i w B_aero[0,0]) (z_hub - z_base)^2 Xi_pitch.
    Returns (L [nrotors,3,6] coefficients of (i w)^p, Gw [nrotors,6,nw] complex or None, (m, hArm) per rotor)."""
    nr = int(fowt.nrotors)
    L = np.zeros((nr, 3, 6))
    Gw = np.zeros((nr, 6, fowt.nw), dtype=complex)
    info = []
    w = np.asarray(fowt.w)
    for ir, rotor in enumerate(fowt.rotorList):
        if only is not None and ir != only:
            continue
        mem_tower = fowt.memberList[fowt.nplatmems + ir]
        if getattr(mem_tower, "type", "rigid") != "rigid":
            raise UnsupportedFOWT("flexible tower: base loads come from the FE stiffness (raft_fowt.py:2540-2601): "
                                  "Engine._save_outputs_general")
        m = fowt.mtower[ir] + rotor.mRNA
        zCG = (fowt.rCG_tow[ir][2] * fowt.mtower[ir] + rotor.r_rel[2] * rotor.mRNA) / m
        zBase = mem_tower.rA[2]
        hArm = zCG - zBase
        r = np.asarray(mem_tower.nodeList[0].r0[:3], dtype=float) - np.array([0.0, 0.0, zCG])
        H = np.array([[0, r[2], -r[1]], [-r[2], 0, r[0]], [r[1], -r[0], 0]])                      # helpers.py:428-437
        Ms = np.asarray(mem_tower.M_struc, dtype=float)
        I44 = (H @ Ms[:3, :3] @ H.T + Ms[3:, :3] @ H + H.T @ Ms[:3, 3:] + Ms[3:, 3:])[1, 1]       # helpers.py:582-583, [4,4]
        ICG = I44 + rotor.mRNA * (rotor.r_rel[2] - zCG) ** 2 + rotor.IrRNA
        L[ir, 0, 4] = m * fowt.g * hArm                       # weight moment
        L[ir, 2, 0] = -m * hArm                               # (i w)^2 = -w^2: inertial reaction, surge part
        L[ir, 2, 4] = -(m * hArm * zCG + ICG)                 # ... and pitch part
        Gw[ir, 4, :] = -(-w ** 2 * fowt.A_aero[0, 0, :, ir] + 1j * w * fowt.B_aero[0, 0, :, ir]) * (rotor.r_rel[2] - zBase) ** 2
        info.append((m, hArm))
    return L, (Gw if np.any(Gw) else None), info


def unit_matrices(fowt, nw):
    """(M0, B0, C0, MBw) of one unit: the sums of raft_model.py:1005-1010,1045-1047."""
    if fowt.nrotors > 0:
        M_turb = np.sum(fowt.A_aero, axis=3)
        B_turb = np.sum(fowt.B_aero, axis=3)
    else:
        M_turb = np.zeros([6, 6, nw])
        B_turb = np.zeros([6, 6, nw])
    A_BEM, B_BEM = np.asarray(fowt.A_BEM), np.asarray(fowt.B_BEM)
    B_gyro = np.sum(fowt.B_gyro, axis=2)
    C_lin = fowt.C_struc + fowt.C_hydro + fowt.C_moor + fowt.C_elast
    if _nonzero(M_turb) or _nonzero(B_turb) or _nonzero(A_BEM) or _nonzero(B_BEM):
        M_lin = M_turb + fowt.M_struc[:, :, None] + A_BEM + fowt.A_hydro_morison[:, :, None]
        B_lin = B_turb + fowt.B_struc[:, :, None] + B_BEM + B_gyro[:, :, None]
        return np.zeros((6, 6)), np.zeros((6, 6)), C_lin, np.array([M_lin, B_lin])
    return fowt.M_struc + fowt.A_hydro_morison, fowt.B_struc + B_gyro, C_lin, None


def sweep_from_models(models, cases, tol=0.01):
    """One batched ``raft_amd.sweep.Sweep`` for the first FOWT of every model (design candidates)
    x every load case (all with the same number of wave headings): what an optimisation driver
    launches instead of ``for model: for case: model.solveDynamics(case)``."""
    from .sweep import Sweep
    f0 = models[0].fowtList[0]
    zeta, beta = [], []
    for case in cases:
        _, b, _, z = waves.sea_state(dict(case), f0.w, f0.dw)
        zeta.append(z)
        beta.append(b)
    rows = []
    for m in models:
        f = m.fowtList[0]
        if int(f.nDOF) != 6:
            raise UnsupportedFOWT("this sweep covers rigid 6-DOF FOWTs (nDOF=%d): units with flexible members go through flex_sweep_from_models" % f.nDOF)
        rows.append((pack_fowt(f),) + unit_matrices(f, m.nw))
    return Sweep.from_fowts(rows, f0.w, f0.k, f0.depth, np.array(zeta), np.array(beta),
                            nIter=int(models[0].nIter), XiStart=models[0].XiStart, tol=tol)


def sweep_from_units(model, cases, tol=0.01):
    """The units of ONE array model as the designs of a Sweep (for ``Sweep.run_farm``): every unit keeps its
    own absolute strip positions, so the wave phase across the farm is carried by the strip table."""
    from .sweep import Sweep
    f0 = model.fowtList[0]
    zeta, beta = [], []
    for case in cases:
        _, b, _, z = waves.sea_state(dict(case), f0.w, f0.dw)
        zeta.append(z)
        beta.append(b)
    rows = [(pack_fowt(f),) + unit_matrices(f, model.nw) for f in model.fowtList]
    return Sweep.from_fowts(rows, f0.w, f0.k, f0.depth, np.array(zeta), np.array(beta),
                            nIter=int(model.nIter), XiStart=model.XiStart, tol=tol)


def sweep_from_member_tables(model, base_table, tables, cases, ctx, tol=0.01, pose=None):
    """A ``raft_amd.sweep.GeometrySweep``: design candidates given as MEMBER DESCRIPTIONS (raft_amd/geometry.py), strip
    tables and statics generated on the device -- no Model()/calcStatics()/calcHydroConstants() per candidate.

    ``model``: a live (reference) Model of the BASE design, positioned and with its statics computed; ``base_table``: the
    MemberTable of that same base design (geometry.describe_unit(design)); ``tables``: DesignTables of the candidates.
    Whatever the generator does not produce -- rotor-nacelle assembly, point inertias, mooring and elastic stiffness,
    structural damping -- is taken from the base model as (reference total) - (generated for the base design), which
    is exact as long as those parts do not change across the candidates (the parametersweep.py / omdao_raft.py case)."""
    from .sweep import GeometrySweep
    from . import geometry as G
    f0 = model.fowtList[0]
    if int(f0.nDOF) != 6:
        raise UnsupportedFOWT("the geometry generator covers rigid 6-DOF FOWTs (nDOF=%d): units with flexible members are built upstream and swept with flex_sweep_from_models" % f0.nDOF)
    M0, B0, C0, MBw = unit_matrices(f0, model.nw)
    if MBw is not None:
        raise UnsupportedFOWT("frequency-dependent base matrices: pass them per design through GeometrySweep(MBw=...)")
    D0 = G.concat_units([base_table])
    Z = np.zeros((1, 6, 6))
    r6 = None if pose is None else np.asarray(pose, dtype=float).reshape(1, 6)
    ctx.build_designs(D0.member_off, D0.members, D0.station_off, D0.stations, Z, Z, Z, model.nw, pose=r6,
                      rho=float(f0.rho_water), g=float(f0.g), k=np.asarray(f0.k), cap_off=D0.cap_off, caps=D0.caps)
    S = ctx.fetch_statics()
    M_extra = M0 - (S["M_struc"][0] + S["A_morison"][0])
    C_extra = C0 - (S["C_struc"][0] + S["C_hydro"][0])
    zeta, beta = [], []
    for case in cases:
        _, b, _, z = waves.sea_state(dict(case), f0.w, f0.dw)
        zeta.append(z)
        beta.append(b)
    nD = tables.n_design
    poses = None if pose is None else np.repeat(r6, nD, axis=0)
    return GeometrySweep(tables, np.repeat(M_extra[None], nD, 0), np.repeat(B0[None], nD, 0), np.repeat(C_extra[None], nD, 0),
                         f0.w, f0.k, f0.depth, np.array(zeta), np.array(beta), nIter=int(model.nIter),
                         XiStart=model.XiStart, tol=tol, pose=poses, rho=float(f0.rho_water), g=float(f0.g))


def flex_sweep_from_models(models, cases, tol=0.01):
    """A ``raft_amd.flex.FlexSweep``: units with MORE than 6 reduced DOFs (flexible members; one per Model, positioned and
    with their statics computed) x load cases in one batch -- the fixed point of raft_model.py:966-1302 for every (unit,
    case) at once, on the device (raftx_flex_solve): node-by-node strip sweeps of the whole batch in one launch per iteration,
    the projections with the units' T, every impedance solve of an iteration in one launch, the convergence test per pair.
    What Engine._solve_general does one case at a time.
    Single strip-theory units only, as there (no potential-flow coefficients, second-order loads, moorMod == 2)."""
    from .flex import FlexSweep, FlexUnit
    eng = Engine(ctx=False)
    units = []
    f0 = models[0].fowtList[0]
    for m in models:
        if len(m.fowtList) != 1 or getattr(m, "ms", None):
            raise UnsupportedFOWT("arrays of units with more than 6 reduced DOFs are not on the device path")
        f = m.fowtList[0]
        if not _general(f):
            raise UnsupportedFOWT("flex_sweep_from_models is for units with more than 6 reduced DOFs (rigid units: sweep_from_units / "
                                  "sweep_from_member_tables)")
        eng._check_supported(f)
        if len(f.w) != len(f0.w) or not np.array_equal(f.w, f0.w):
            raise UnsupportedFOWT("the units of a flexible sweep must share their frequency grid")
        units.append(FlexUnit.from_fowt(f))
    zeta, beta = [], []
    for case in cases:
        _, b, _, z = waves.sea_state(dict(case), f0.w, f0.dw)
        zeta.append(z)
        beta.append(b)
    return FlexSweep(units, f0.w, f0.k, f0.depth, np.array(zeta), np.array(beta), nIter=int(models[0].nIter),
                     XiStart=models[0].XiStart, tol=tol)


_default_engine = Engine()


def use_context(ctx):
    """Bind the module-level drop-ins to an explicit raftx context (tests use
    this to drive the same host code against the CPU oracle)."""
    global _default_engine
    _default_engine = Engine(ctx)
    return _default_engine


def calcHydroExcitation(fowt, case, memberList=[]):
    return _default_engine.calcHydroExcitation(fowt, case, memberList)


def calcHydroLinearization(fowt, Xi):
    return _default_engine.calcHydroLinearization(fowt, Xi)


def calcDragExcitation(fowt, ih):
    return _default_engine.calcDragExcitation(fowt, ih)


def calcQTF_slenderBody(fowt, waveHeadInd, Xi0=None, verbose=False, iCase=None, iWT=None):
    return _default_engine.calcQTF_slenderBody(fowt, waveHeadInd, Xi0=Xi0, verbose=verbose, iCase=iCase, iWT=iWT)


def calcHydroForce_2ndOrd(fowt, beta, S0, iCase=None, iWT=None, interpMode='qtf'):
    return _default_engine.calcHydroForce_2ndOrd(fowt, beta, S0, iCase=iCase, iWT=iWT, interpMode=interpMode)


def saveTurbineOutputs(fowt, results, case):
    return _default_engine.saveTurbineOutputs(fowt, results, case)


def solveDynamics(model, case, tol=0.01, conv_plot=0, RAO_plot=0, display=0):
    return _default_engine.solveDynamics(model, case, tol=tol, conv_plot=conv_plot,
                                         RAO_plot=RAO_plot, display=display)


def install(raft_module=None, outputs=False, materialise_members=None):
    """Monkey-patch a loaded reference package so that Model.analyzeCases & co
    run the hot path on the GPU.  Returns the originals for un-patching.
    outputs=True also routes FOWT.saveTurbineOutputs (statistics of the resident responses; rigid single units
    without MoorPy / controller outputs -- anything else raises UnsupportedFOWT, never a silent fallback).
    materialise_members=True / False: switch the default engine's per-member side effects (mem.u, ud, pDyn,
    F_hydro_iner, Bmat, F_exc_drag: Engine.materialise_members) on / off; None leaves the engine as it is."""
    if materialise_members is not None:
        _default_engine.materialise_members = bool(materialise_members)
    if raft_module is None:
        import raft as raft_module
    from raft import raft_model, raft_fowt
    saved = dict(solveDynamics=raft_model.Model.solveDynamics,
                 calcHydroExcitation=raft_fowt.FOWT.calcHydroExcitation,
                 calcHydroLinearization=raft_fowt.FOWT.calcHydroLinearization,
                 calcDragExcitation=raft_fowt.FOWT.calcDragExcitation,
                 calcQTF_slenderBody=raft_fowt.FOWT.calcQTF_slenderBody,
                 calcHydroForce_2ndOrd=raft_fowt.FOWT.calcHydroForce_2ndOrd)
    raft_model.Model.solveDynamics = solveDynamics
    raft_fowt.FOWT.calcHydroExcitation = calcHydroExcitation
    raft_fowt.FOWT.calcHydroLinearization = calcHydroLinearization
    raft_fowt.FOWT.calcDragExcitation = calcDragExcitation
    raft_fowt.FOWT.calcQTF_slenderBody = calcQTF_slenderBody
    raft_fowt.FOWT.calcHydroForce_2ndOrd = calcHydroForce_2ndOrd
    if outputs:
        saved["saveTurbineOutputs"] = raft_fowt.FOWT.saveTurbineOutputs
        raft_fowt.FOWT.saveTurbineOutputs = saveTurbineOutputs
    return saved


def uninstall(saved):
    from raft import raft_model, raft_fowt
    raft_model.Model.solveDynamics = saved['solveDynamics']
    raft_fowt.FOWT.calcHydroExcitation = saved['calcHydroExcitation']
    raft_fowt.FOWT.calcHydroLinearization = saved['calcHydroLinearization']
    raft_fowt.FOWT.calcDragExcitation = saved['calcDragExcitation']
    raft_fowt.FOWT.calcQTF_slenderBody = saved['calcQTF_slenderBody']
    raft_fowt.FOWT.calcHydroForce_2ndOrd = saved['calcHydroForce_2ndOrd']
    if 'saveTurbineOutputs' in saved:
        raft_fowt.FOWT.saveTurbineOutputs = saved['saveTurbineOutputs']
