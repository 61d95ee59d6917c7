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

Not covered by the device path (raises, never falls back silently):
flexible / >6-DOF FOWTs, moorMod==2 per-iteration mooring damping,
potSecOrder==1 (internal slender-body QTF re-entry), submerged rotors.
Per-member intermediates (mem.u, mem.ud, mem.pDyn, mem.Bmat, mem.F_exc_drag)
are consumed only inside the replaced methods and are not materialised.
"""
import numpy as np

from . import waves
from .strips import pack_fowt, UnsupportedFOWT
from . import backend


class Engine:
    """Binds the host mirror to one raftx context (default: the HIP library)."""

    def __init__(self, ctx=None):
        self._ctx = ctx

    @property
    def ctx(self):
        if self._ctx is None:
            self._ctx = backend.default_context()
        return self._ctx

    # ------------------------------------------------------------------
    def _sea_state(self, fowt, case):
        nWaves, beta, S, zeta = waves.sea_state(case, fowt.w, fowt.dw)
        fowt.nWaves, fowt.beta, fowt.S, fowt.zeta = nWaves, beta, S, zeta

    def _check_supported(self, fowt):
        if int(fowt.nDOF) != 6:
            raise UnsupportedFOWT("device path covers rigid 6-DOF FOWTs (nDOF=%d)" % fowt.nDOF)
        for rot in getattr(fowt, "rotorList", []):
            if rot.r3[2] < 0:
                raise UnsupportedFOWT("submerged rotors (raft_fowt.py:1861-1883) are not on the device path")

    def _F_BEM(self, fowt, case):
        """raft_fowt.py:1788-1849,1887 -- potential-flow excitation with heading
        interpolation (host; these are dense per-bin inputs to the solve)."""
        nw = fowt.nw
        F_full = np.zeros([fowt.nWaves, fowt.nFullDOF, nw], dtype=complex)
        if getattr(fowt, "potMod", False) or getattr(fowt, "potModMaster", 1) in [2, 3]:
            for ih in range(fowt.nWaves):
                hd = np.deg2rad(case['wave_heading'][ih])
                phase_offset = np.exp(-1j * fowt.k * (fowt.x_ref * np.cos(hd) + fowt.y_ref * np.sin(hd)))
                beta = (np.degrees(fowt.beta[ih]) - fowt.heading_adjust) % 360
                headings = fowt.BEM_headings
                nhs = len(headings)
                if beta <= headings[0]:
                    hlast = headings[-1] - 360
                    i1, i2 = nhs - 1, 0
                    f2 = (beta - hlast) / (headings[0] - hlast)
                elif beta >= headings[nhs - 1]:
                    hfirst = headings[0] + 360
                    i1, i2 = nhs - 1, 0
                    f2 = (beta - headings[-1]) / (hfirst - headings[-1])
                else:
                    for i in range(nhs - 1):
                        if headings[i + 1] > beta:
                            i1, i2 = i, i + 1
                            f2 = (beta - headings[i]) / (headings[i + 1] - headings[i])
                            break
                f1 = 1.0 - f2
                Xp = fowt.X_BEM[i1, :, :] * f1 + fowt.X_BEM[i2, :, :] * f2
                sb, cb = np.sin(fowt.beta[ih]), np.cos(fowt.beta[ih])
                X = np.zeros([6, nw], dtype=complex)
                X[0, :] = Xp[0, :] * cb - Xp[1, :] * sb
                X[1, :] = Xp[0, :] * sb + Xp[1, :] * cb
                X[2, :] = Xp[2, :]
                X[3, :] = Xp[3, :] * cb - Xp[4, :] * sb
                X[4, :] = Xp[3, :] * sb + Xp[4, :] * cb
                X[5, :] = Xp[5, :]
                F_full[ih, :6, :] = X * fowt.zeta[ih, :] * phase_offset
        F = np.zeros([fowt.nWaves, fowt.nDOF, nw], dtype=complex)
        for ih in range(fowt.nWaves):
            F[ih] = fowt.T.T @ F_full[ih]
        return F, F_full

    def _upload(self, fowts, case_zeta, case_beta, mats=None):
        """Upload N units + one sea state.  mats: per-unit (M0,B0,C0,MBw)."""
        f0 = fowts[0]
        nw = len(f0.w)
        tables = [f._raftx_table for f in fowts]
        nD = len(fowts)
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
        ctx.upload_cases(f0.w, f0.k, f0.depth, 1025.0, 9.81,
                         np.asarray(case_zeta)[None, :, :], np.asarray(case_beta)[None, :])

    # ------------------------------------------------------------------
    def calcHydroExcitation(self, fowt, case, memberList=[]):
        """raft_fowt.py:1732-1888."""
        self._check_supported(fowt)
        self._sea_state(fowt, case)
        fowt.F_BEM, fowt.F_BEM_fullDOF = self._F_BEM(fowt, case)
        fowt._raftx_table = pack_fowt(fowt, memberList if len(memberList) else None)
        self._upload([fowt], fowt.zeta, fowt.beta)
        F = self.ctx.excitation()[0, 0]                     # [nWaves,6,nw]
        fowt.F_hydro_iner = F
        fowt._raftx_fresh = True
        return None

    def calcHydroLinearization(self, fowt, Xi):
        """raft_fowt.py:1891-1936 (heading 0 only, :1910)."""
        if not hasattr(fowt, "_raftx_table"):
            raise RuntimeError("calcHydroExcitation must be called before calcHydroLinearization")
        self._upload([fowt], fowt.zeta, fowt.beta)
        B, F = self.ctx.linearize(np.asarray(Xi, dtype=complex)[None, None, :, :])
        fowt.B_hydro_drag = B[0, 0]
        fowt._raftx_Fdrag = F[0, 0]                         # [nWaves,6,nw]
        fowt.F_hydro_drag = F[0, 0, 0].copy()
        return fowt.B_hydro_drag

    def calcDragExcitation(self, fowt, ih):
        """raft_fowt.py:1940-1957."""
        if not hasattr(fowt, "_raftx_Fdrag"):
            raise RuntimeError("calcHydroLinearization must be called before calcDragExcitation")
        fowt.F_hydro_drag = fowt._raftx_Fdrag[ih].copy()
        return fowt.F_hydro_drag

    # ------------------------------------------------------------------
    def solveDynamics(self, model, case, tol=0.01, conv_plot=0, RAO_plot=0, display=0):
        """raft_model.py:966-1302."""
        iCase = case['iCase'] if 'iCase' in case else None
        fowts = model.fowtList
        nF = len(fowts)
        nw = model.nw
        mats, F_extras = [], []
        for i, fowt in enumerate(fowts):
            self._check_supported(fowt)
            if getattr(fowt, "ms", None) and getattr(fowt, "moorMod", 0) == 2:
                raise UnsupportedFOWT("moorMod==2 (raft_model.py:1023-1030,1069-1072) is not on the device path")
            if getattr(fowt, "potSecOrder", 0) == 1:
                raise UnsupportedFOWT("potSecOrder==1 (internal QTF re-entry, raft_model.py:1108-1131) "
                                      "is not on the device path yet")
            # sea state + excitation inputs (raft_model.py:1002)
            self._sea_state(fowt, case)
            fowt.F_BEM, fowt.F_BEM_fullDOF = self._F_BEM(fowt, case)
            fowt._raftx_table = pack_fowt(fowt)

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
                    fowt.calcHydroForce_2ndOrd(fowt.beta[0], fowt.S[0, :], iCase=iCase, iWT=i)
                for ih in range(1, fowt.nWaves):                            # :1210-1211
                    fowt.Fhydro_2nd_mean[ih, :], fowt.Fhydro_2nd[ih, :, :] = \
                        fowt.calcHydroForce_2ndOrd(fowt.beta[ih], fowt.S[ih, :])
            C_moor = fowt.C_moor
            A_BEM = np.asarray(fowt.A_BEM)
            B_BEM = np.asarray(fowt.B_BEM)
            B_gyro = np.sum(fowt.B_gyro, axis=2)
            C_lin = fowt.C_struc + fowt.C_hydro + C_moor + fowt.C_elast    # :1047
            if np.any(M_turb) or np.any(B_turb) or np.any(A_BEM) or np.any(B_BEM):
                M_lin = M_turb + fowt.M_struc[:, :, None] + A_BEM + fowt.A_hydro_morison[:, :, None]   # :1045
                B_lin = B_turb + fowt.B_struc[:, :, None] + B_BEM + B_gyro[:, :, None]                 # :1046
                mats.append((np.zeros((6, 6)), np.zeros((6, 6)), C_lin, np.array([M_lin, B_lin])))
            else:
                mats.append((fowt.M_struc + fowt.A_hydro_morison, fowt.B_struc + B_gyro, C_lin, None))
            F_extras.append(fowt.F_BEM + fowt.Fhydro_2nd)

        f0 = fowts[0]
        self._upload(fowts, f0.zeta, f0.beta, mats)
        ctx = self.ctx
        F_extra = np.array(F_extras)[:, None]                               # [nF,1,nH,6,nw]
        F_iner = ctx.excitation()                                            # side effect of :1002
        out = ctx.solve_dynamics(int(model.nIter), tol=tol, XiStart=model.XiStart,
                                 F_extra=F_extra if np.any(F_extra) else None,
                                 want_Xi=True, want_B=True, want_F=True, want_Z=True)
        if np.any(out['flags'] & 2):
            raise Exception("Nan detected in response vector Xi.")          # :1098-1099
        nH = f0.nWaves
        for i, fowt in enumerate(fowts):
            fowt.F_hydro_iner = F_iner[i, 0]
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
            Cc = None
            if ms:                                                          # :1173-1182
                if getattr(model, "moorMod", 0) in (0, 1):
                    Cc = np.asarray(ms.getCoupledStiffnessA(lines_only=True), dtype=float)[None]
                else:
                    raise UnsupportedFOWT("array-level moorMod==2 is not on the device path")
            Zblk = out['Z'][:, 0][None]                                     # [1,nF,6,6,nw]
            Fw = np.transpose(out['F_wave'][:, 0], (1, 0, 2, 3)).reshape(1, nH, n, nw)
            model.Xi[:nH] = ctx.solve_system(model.w, Zblk, Fw, Cc=Cc)[0]
        for i, fowt in enumerate(fowts):                                    # :1251-1255
            fowt.Xi = model.Xi[:, i * fowt.nDOF:(i + 1) * fowt.nDOF, :]
            fowt.Xi_fullDOF = np.zeros([fowt.nWaves + 1, fowt.nFullDOF, nw], dtype=complex)
            for ih in range(fowt.nWaves + 1):
                fowt.Xi_fullDOF[ih, :, :] = fowt.T @ fowt.Xi[ih, :, :]
        model.results['response'] = {}                                      # :1300
        model._raftx_niter = out['niter'][:, 0].copy()
        model._raftx_flags = out['flags'][:, 0].copy()
        return model.Xi


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
    if np.any(M_turb) or np.any(B_turb) or np.any(A_BEM) or np.any(B_BEM):
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
            raise UnsupportedFOWT("device path covers rigid 6-DOF FOWTs (nDOF=%d)" % f.nDOF)
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


def solveDynamics(model, case, tol=0.01, conv_plot=0, RAO_plot=0, display=0):
    return _default_engine.solveDynamics(model, case, tol=tol, conv_plot=conv_plot,
                                         RAO_plot=RAO_plot, display=display)


def install(raft_module=None):
    """Monkey-patch a loaded reference package so that Model.analyzeCases & co
    run the hot path on the GPU.  Returns the originals for un-patching."""
    if raft_module is None:
        import raft as raft_module
    from raft import raft_model, raft_fowt
    saved = dict(solveDynamics=raft_model.Model.solveDynamics,
                 calcHydroExcitation=raft_fowt.FOWT.calcHydroExcitation,
                 calcHydroLinearization=raft_fowt.FOWT.calcHydroLinearization,
                 calcDragExcitation=raft_fowt.FOWT.calcDragExcitation)
    raft_model.Model.solveDynamics = solveDynamics
    raft_fowt.FOWT.calcHydroExcitation = calcHydroExcitation
    raft_fowt.FOWT.calcHydroLinearization = calcHydroLinearization
    raft_fowt.FOWT.calcDragExcitation = calcDragExcitation
    return saved


def uninstall(saved):
    from raft import raft_model, raft_fowt
    raft_model.Model.solveDynamics = saved['solveDynamics']
    raft_fowt.FOWT.calcHydroExcitation = saved['calcHydroExcitation']
    raft_fowt.FOWT.calcHydroLinearization = saved['calcHydroLinearization']
    raft_fowt.FOWT.calcDragExcitation = saved['calcDragExcitation']
