"""Strip-table packer: the host-side feeder of the hot path.

Walks a FOWT's ``memberList`` (the reference's objects, or any duck-typed
stand-in exposing the same attributes) and emits one 32-double record per
*submerged* Morison strip -- the "LDS-staged member geometry" the kernels
consume -- plus the optional MacCamy-Fuchs complex Cm table.

It restates, on the host, the per-strip constants of
    raft/raft_member.py:1261-1368  Member.calcHydroConstants  (a_i, volumes)
    raft/raft_member.py:1370-1448  Member.calcImat            (Imat scalars)
    raft/raft_member.py:1451-1486  Member.getCmSides          (Cm, MCF ramp)
    raft/raft_member.py:2058-2110  drag areas / coefficients of
                                   Member.calcHydroLinearization
and folds the member-node -> reduced-DOF map (raft/raft_fowt.py:1919-1929,
node.T) into a single arm per strip, which is exact for rigid 6-DOF FOWTs
(SURVEY.md Appendix A).

Record layout (doubles; see include/raftx.h RAFTX_F_*):
   0..2   x y z        absolute strip position (wave phase, depth decay)
   3..5   ax ay az     arm from the reduced-DOF reference point
   6..8   q            axial unit vector
   9..11  p1           transverse unit vector 1
  12..14  p2           transverse unit vector 2
  15      Iq           rho*v_end*Ca_End
  16      Ip1          rho*v_side*Cm_p1   (0 when the strip uses the MCF table)
  17      Ip2          rho*v_side*Cm_p2   (0 when the strip uses the MCF table)
  18      a_i          signed end area for the dynamic-pressure force
  19..22  dq dp1 dp2 dEnd   sqrt(8/pi)*0.5*rho*area*Cd  (times vRMS on device)
  23      circ         1.0 circular (total transverse vRMS), 0.0 rectangular
  24      mcf          -1.0, or row index into the complex Cm table
  25      rhoV         rho*v_side (multiplies the complex Cm of the MCF table)
  26      member index (diagnostic)
  27      strip index within member (diagnostic)
  28      run hint STEP: this strip = previous strip + STEP*UNIT*q (1..4), 0 = run start
  29      run hint UNIT: smallest strip spacing of the member [m]
  30..31  reserved (0)
The run hints let the device advance the wave kinematics along a member with
rotors instead of re-evaluating sincos/exp per strip; they are verified against
x,y,z when uploaded (include/raftx.h) and ignored by the oracle.
"""
import numpy as np

NFIELD = 32
(F_X, F_Y, F_Z, F_AX, F_AY, F_AZ) = range(6)
F_Q, F_P1, F_P2 = 6, 9, 12
F_IQ, F_IP1, F_IP2, F_AI = 15, 16, 17, 18
F_DQ, F_DP1, F_DP2, F_DEND = 19, 20, 21, 22
F_CIRC, F_MCF, F_RHOV, F_MEM, F_IL = 23, 24, 25, 26, 27
F_STEP, F_UNIT = 28, 29
MAX_RUN = 16


class UnsupportedFOWT(Exception):
    """Raised when a FOWT is outside what the device path covers; callers fall
    back loudly, never silently."""


def node_arm(T):
    """Arm (r_node - rP) encoded in a rigid node's 6x6 map T (raft_node.py:262-290):
    T = [[I, H(arm)], [0, I]] with H as raft/helpers.py:428-437."""
    T = np.asarray(T, dtype=float)
    if T.shape != (6, 6):
        raise UnsupportedFOWT("node.T is %s, expected (6, 6): not a rigid 6-DOF FOWT" % (T.shape,))
    I3 = np.eye(3)
    if not (np.allclose(T[:3, :3], I3, atol=1e-12) and np.allclose(T[3:, 3:], I3, atol=1e-12)
            and np.allclose(T[3:, :3], 0.0, atol=1e-12)):
        raise UnsupportedFOWT("node.T is not a rigid-body translation map")
    arm = np.array([T[1, 5], T[2, 3], T[0, 4]])
    H = np.array([[0, arm[2], -arm[1]], [-arm[2], 0, arm[0]], [arm[1], -arm[0], 0]])
    if not np.allclose(T[:3, 3:], H, atol=1e-9):
        raise UnsupportedFOWT("node.T translation block is not an alternator matrix")
    return arm


def _interp(x, xp, fp):
    return float(np.interp(x, xp, fp))


def cm_sides(mem, il, k):
    """Complex (Cm_p1, Cm_p2) of strip ``il`` at wave number ``k`` with the
    MacCamy-Fuchs correction and its cosine ramp: raft_member.py:1459-1484."""
    from scipy.special import hankel1
    Ca_p1 = _interp(mem.ls[il], mem.stations, mem.Ca_p1)
    Ca_p2 = _interp(mem.ls[il], mem.stations, mem.Ca_p2)
    Cm1_0, Cm2_0 = 1.0 + Ca_p1, 1.0 + Ca_p2
    R = mem.ds[il] / 2
    Hp1 = 0.5 * (hankel1(0, k * R) - hankel1(2, k * R))
    Cm = 4j / (np.pi * (k * R) ** 2 * Hp1)
    Tr = np.pi / 5 / R
    T0 = 0
    ramp = 0.5 * (1 - np.cos(np.pi * (k - T0) / Tr)) if k < Tr else 1
    ramp = 0 if k <= T0 else ramp
    return Cm * ramp + Cm1_0 * (1 - ramp), Cm * ramp + Cm2_0 * (1 - ramp)


def cm_sides_array(mem, il, k):
    """cm_sides for a whole wave-number array at once (raft_member.py:1459-1484): [2,nk] complex."""
    from scipy.special import hankel1
    k = np.asarray(k, dtype=float)
    Ca_p1 = _interp(mem.ls[il], mem.stations, mem.Ca_p1)
    Ca_p2 = _interp(mem.ls[il], mem.stations, mem.Ca_p2)
    R = mem.ds[il] / 2
    with np.errstate(divide="ignore", invalid="ignore"):
        Hp1 = 0.5 * (hankel1(0, k * R) - hankel1(2, k * R))
        Cm = 4j / (np.pi * (k * R) ** 2 * Hp1)
    Tr = np.pi / 5 / R
    ramp = np.where(k < Tr, 0.5 * (1 - np.cos(np.pi * k / Tr)), 1.0)
    ramp = np.where(k <= 0, 0.0, ramp)
    Cm = np.where(ramp == 0.0, 0.0, Cm)                       # k <= 0: the ramp removes the (singular) MCF value
    return np.array([Cm * ramp + (1.0 + Ca_p1) * (1 - ramp), Cm * ramp + (1.0 + Ca_p2) * (1 - ramp)])


def pack_member(mem, imem, rho, k_array=None, arm_node=None):
    """Records for the submerged strips of one rigid member (vectorised over the member's strips).

    Returns (records [n,32], cm rows list of complex [2,nw])."""
    rigid = getattr(mem, "type", "rigid") == "rigid"
    if not rigid and arm_node is None:
        raise UnsupportedFOWT("member '%s' is type '%s': a flexible member has no single arm to the reduced-DOF point "
                              "(pack_fowt_nodes gives its strips node by node)" % (getattr(mem, "name", "?"), mem.type))
    circ = (mem.shape == "circular")
    potMod = bool(getattr(mem, "potMod", False))
    MCF = bool(getattr(mem, "MCF", False)) and circ and (k_array is not None)
    node = mem.nodeList[0]
    if arm_node is None:
        arm_node = node_arm(node.T)
    if rigid:
        r_node = np.asarray(node.r, dtype=float)[:3]
    else:                                  # strip il hangs on structural node il (raft_member.py:1969-1976, 2051-2056)
        if len(mem.nodeList) != int(mem.ns):
            raise UnsupportedFOWT("flexible member '%s': %d structural nodes for %d strips"
                                  % (getattr(mem, "name", "?"), len(mem.nodeList), mem.ns))
        r_node = np.array([np.asarray(nd.r, dtype=float)[:3] for nd in mem.nodeList])
    q = np.asarray(mem.q, dtype=float)
    p1 = np.asarray(mem.p1, dtype=float)
    p2 = np.asarray(mem.p2, dtype=float)
    c_drag = np.sqrt(8 / np.pi)

    r_all = np.asarray(mem.r, dtype=float).reshape(-1, 3)
    wet = np.nonzero(r_all[:, 2] < 0)[0]                      # raft_member.py:1979,2058
    n = len(wet)
    if n == 0:
        return np.zeros((0, NFIELD)), []
    ls = np.asarray(mem.ls, dtype=float)[wet]
    dls = np.asarray(mem.dls, dtype=float)[wet]
    r = r_all[wet]
    stations = np.asarray(mem.stations, dtype=float)
    coef = lambda name: np.interp(ls, stations, np.asarray(getattr(mem, name), dtype=float))
    rec = np.zeros((n, NFIELD))

    # run hints (optional: the library re-derives and verifies them at upload)
    consecutive = np.diff(wet) == 1
    gaps = np.diff(ls)[consecutive]
    gaps = gaps[gaps > 0]
    unit = float(gaps.min()) if len(gaps) else 0.0
    run = 0
    for i in range(1, n):
        step = 0
        if consecutive[i - 1] and unit > 0 and run < MAX_RUN:
            ratio = (ls[i] - ls[i - 1]) / unit
            m = int(round(ratio))
            if 1 <= m <= 4 and abs(ratio - m) < 1e-9:
                step = m
        run = run + 1 if step else 0
        rec[i, F_STEP] = step
    rec[:, F_UNIT] = unit
    rec[:, F_X:F_X + 3] = r
    rec[:, F_AX:F_AX + 3] = (r - (r_node if rigid else r_node[wet])) + arm_node
    rec[:, F_Q:F_Q + 3] = q
    rec[:, F_P1:F_P1 + 3] = p1
    rec[:, F_P2:F_P2 + 3] = p2
    rec[:, F_CIRC] = 1.0 if circ else 0.0
    rec[:, F_MCF] = -1.0
    rec[:, F_MEM] = imem
    rec[:, F_IL] = wet
    ds = np.asarray(mem.ds, dtype=float)[wet]
    drs = np.asarray(mem.drs, dtype=float)[wet]

    # ---- inertial-excitation scalars (raft_member.py:1395-1448, :1340-1348)
    cms = []
    if not potMod:
        if circ:
            v_i = 0.25 * np.pi * ds ** 2 * dls
            v_end = np.pi / 12.0 * np.abs((ds + drs) ** 3 - (ds - drs) ** 3)
            a_i = np.pi * ds * drs
        else:
            v_i = ds[:, 0] * ds[:, 1] * dls
            v_end = np.pi / 12.0 * (np.mean(ds + drs, axis=1) ** 3 - np.mean(ds - drs, axis=1) ** 3)
            a_i = (ds[:, 0] + drs[:, 0]) * (ds[:, 1] + drs[:, 1]) - (ds[:, 0] - drs[:, 0]) * (ds[:, 1] - drs[:, 1])
        pierce = r[:, 2] + 0.5 * dls > 0                      # strip pierces the waterline
        with np.errstate(divide="ignore", invalid="ignore"):
            v_i = np.where(pierce, v_i * (0.5 * dls - r[:, 2]) / dls, v_i)
        rec[:, F_IQ] = rho * v_end * coef("Ca_End")
        rec[:, F_AI] = a_i
        rec[:, F_RHOV] = rho * v_i
        if MCF:
            for i, il in enumerate(wet):
                rec[i, F_MCF] = float(len(cms))
                cms.append(cm_sides_array(mem, il, k_array))
        else:
            rec[:, F_IP1] = rho * v_i * (1.0 + coef("Ca_p1"))
            rec[:, F_IP2] = rho * v_i * (1.0 + coef("Ca_p2"))

    # ---- drag scalars (raft_member.py:2061-2110); note the reference's
    # rectangular axial area 2*(ds0+ds0)*dl (sic, :2070)
    if circ:
        a_q = np.pi * ds * dls
        a_p1 = ds * dls
        a_p2 = ds * dls
        a_end = np.abs(np.pi * ds * drs)
    else:
        a_q = 2 * (ds[:, 0] + ds[:, 0]) * dls
        a_p1 = ds[:, 0] * dls
        a_p2 = ds[:, 1] * dls
        a_end = np.abs((ds[:, 0] + drs[:, 0]) * (ds[:, 1] + drs[:, 1]) - (ds[:, 0] - drs[:, 0]) * (ds[:, 1] - drs[:, 1]))
    rec[:, F_DQ] = c_drag * 0.5 * rho * a_q * coef("Cd_q")
    rec[:, F_DP1] = c_drag * 0.5 * rho * a_p1 * coef("Cd_p1")
    rec[:, F_DP2] = c_drag * 0.5 * rho * a_p2 * coef("Cd_p2")
    rec[:, F_DEND] = c_drag * 0.5 * rho * a_end * coef("Cd_End")
    return rec, cms


class StripTable:
    """Packed submerged strips of one FOWT ("design")."""

    def __init__(self, strips, cm_mcf=None):
        self.strips = np.ascontiguousarray(strips, dtype=np.float64).reshape(-1, NFIELD)
        self.cm_mcf = None if cm_mcf is None or len(cm_mcf) == 0 else \
            np.ascontiguousarray(cm_mcf, dtype=np.complex128)

    @property
    def n(self):
        return self.strips.shape[0]


_PACK_INPUTS = ("r", "q", "p1", "p2", "ds", "drs", "dls", "ls", "stations", "Cd_q", "Cd_p1", "Cd_p2", "Cd_End",
                "Ca_p1", "Ca_p2", "Ca_End")


def pack_fingerprint(fowt, memberList=None):
    """Bytes of everything ``pack_fowt`` reads: two calls with equal fingerprints pack identical tables.  A caller that
    solves many load cases on one unit (Model.analyzeCases) re-packs only when a member moved or was edited -- the
    fingerprint costs a tenth of the packing."""
    members = fowt.memberList if memberList is None else memberList
    parts = [np.float64(fowt.rho_water).tobytes()]
    for mem in members:
        node = mem.nodeList[0]
        parts.append(("%s|%s|%d|%d" % (getattr(mem, "type", "rigid"), mem.shape, bool(getattr(mem, "potMod", False)),
                                       bool(getattr(mem, "MCF", False)))).encode())
        parts.append(np.asarray(node.r, dtype=float).tobytes())
        parts.append(np.asarray(node.T, dtype=float).tobytes())
        for a in _PACK_INPUTS:
            parts.append(np.asarray(getattr(mem, a), dtype=float).tobytes())
        if getattr(mem, "MCF", False):
            parts.append(np.asarray(fowt.k, dtype=float).tobytes())
    return b"".join(parts)


def pack_fowt(fowt, memberList=None, own_node=False):
    """StripTable for a rigid 6-DOF FOWT (reference object or stand-in).  memberList None: every member of the unit;
    a list (possibly empty): exactly those.  own_node: arms about each member's own node instead of the unit's
    reduced-DOF point (the per-member vectors of F_hydro_iner_fullDOF, raft_fowt.py:1853-1857)."""
    if int(getattr(fowt, "nDOF", 6)) != 6:
        raise UnsupportedFOWT("FOWT has %d reduced DOFs; the device path covers rigid 6-DOF units"
                              % fowt.nDOF)
    members = fowt.memberList if memberList is None else memberList
    rho = float(fowt.rho_water)
    recs, cms = [], []
    for imem, mem in enumerate(members):
        k_array = np.asarray(fowt.k, dtype=float) if getattr(mem, "MCF", False) else None
        r, c = pack_member(mem, imem, rho, k_array=k_array, arm_node=np.zeros(3) if own_node else None)
        if len(c):
            r = r.copy()
            sel = r[:, F_MCF] >= 0
            r[sel, F_MCF] += len(cms)
            cms.extend(c)
        recs.append(r)
    strips = np.concatenate(recs, axis=0) if recs else np.zeros((0, NFIELD))
    return StripTable(strips, np.array(cms) if cms else None)


def pack_fowt_nodes(fowt, memberList=None):
    """The strips of a unit grouped by the STRUCTURAL NODE that carries them: (rows, tables) with rows[i] the first
    full-DOF row (node.id * 6) of table i, arms about that node's own position.  A rigid member is one table on its
    single node; a flexible member gives one single-strip table per wet node (raft_member.py:1969-1976, 2046-2056).
    These tables are the "designs" of raftx_excitation / raftx_linearize for a unit with more than 6 reduced DOFs:
    their 6-vectors and 6 x 6 blocks are rows / diagonal blocks of the full-DOF arrays that the unit's T matrix reduces
    (raft_fowt.py:1853-1857, 1912-1929)."""
    members = fowt.memberList if memberList is None else memberList
    rho = float(fowt.rho_water)
    rows, tables = [], []
    for imem, mem in enumerate(members):
        mcf = bool(getattr(mem, "MCF", False))
        k_array = np.asarray(fowt.k, dtype=float) if mcf else None
        rec, cms = pack_member(mem, imem, rho, k_array=k_array, arm_node=np.zeros(3))
        if len(rec) == 0:
            continue
        if getattr(mem, "type", "rigid") == "rigid":
            node = mem.nodeList[0]
            rows.append(int(node.id) * int(getattr(node, "nDOF", 6)))
            tables.append(StripTable(rec, np.array(cms) if cms else None))
            continue
        for one in rec:
            node = mem.nodeList[int(one[F_IL])]
            one = one.copy()
            one[F_STEP] = one[F_UNIT] = 0.0                  # a table of one strip has no run
            cm_one = None
            if mcf and one[F_MCF] >= 0:                      # MacCamy-Fuchs on a flexible member (raft_member.py:1415-1420, 1969-1976):
                cm_one = np.array(cms[int(one[F_MCF])])[None]    # the strip's own row of the Cm table goes with its table
                one[F_MCF] = 0.0
            rows.append(int(node.id) * int(getattr(node, "nDOF", 6)))
            tables.append(StripTable(one[None, :], cm_one))
    return rows, tables


# ---------------------------------------------------------------------------- submerged rotors (raft_fowt.py:1861-1883)
def _sym_couple_basis():
    """Least-norm map from a general 3x3 matrix M to three symmetric matrices S_x, S_y, S_z with
    M = [e_x]x S_x + [e_y]x S_y + [e_z]x S_z  ([d]x = cross-product matrix): 9 equations, 18 unknowns."""
    idx = [(0, 0), (0, 1), (0, 2), (1, 1), (1, 2), (2, 2)]
    A = np.zeros((9, 18))
    for kk in range(3):
        e = np.zeros(3)
        e[kk] = 1.0
        X = np.array([[0, -e[2], e[1]], [e[2], 0, -e[0]], [-e[1], e[0], 0]])
        for j, (a, b) in enumerate(idx):
            S = np.zeros((3, 3))
            S[a, b] = S[b, a] = 1.0
            A[:, kk * 6 + j] = (X @ S).reshape(9)
    return np.linalg.pinv(A), idx


_COUPLE_PINV, _SYM_IDX = _sym_couple_basis()


def pack_rotors(fowt, with_node_arm, only=None):
    """Pseudo-strips that make the device's inertial-excitation sweep produce the force of every SUBMERGED rotor
    (raft_fowt.py:1861-1883):  f3 = I3 ud,  f6 = [f3 ; a x f3 + M3 ud]  with I_hydro = rotateMatrix6(rot.I_hydro, rot.R_q),
    I3 = I_hydro[:3,:3], M3 = I_hydro[3:,:3], ud the wave acceleration at the hub and a = rot.r3 - r6[:3].
    The strip form is  f3 = sum_c I_c n_c (n_c . ud)  at an arm: the symmetric I3 is one strip (eigenvectors as its triad,
    eigenvalues as its inertia scalars); the moment block M3 is written as  sum_k [e_k]x S_k  with symmetric S_k (least
    norm) and every term is a couple -- two strips at the hub (same kinematics) with arms a +- e_k / 2 and inertia
    +-S_k: no net force, moment e_k x (S_k ud).  with_node_arm: add the arm of the rotor's node to the reduced-DOF point
    (what T^T applied to the rotor node's slots of the full-DOF vector adds, :1886-1888); without it the table gives the
    vector about the PRP that goes into those slots.  only: index into rotorList of the ONE rotor to pack (default: all
    submerged rotors in one table -- their forces add).  Returns a StripTable, or None if no rotor is submerged."""
    rows = []
    for ir, rot in enumerate(getattr(fowt, "rotorList", [])):
        r3 = np.asarray(rot.r3, dtype=float)
        if not r3[2] < 0 or (only is not None and ir != only):
            continue
        R = np.asarray(rot.R_q, dtype=float)
        I6 = np.asarray(rot.I_hydro, dtype=float)
        I3 = R @ I6[:3, :3] @ R.T                               # rotateMatrix6 (helpers.py:604-629): mass block ...
        M3 = (R @ I6[:3, 3:] @ R.T).T                           # ... and the transposed product-of-inertia block
        if not np.allclose(I3, I3.T, rtol=1e-9, atol=1e-9 * max(np.abs(I3).max(), 1e-300)):
            raise UnsupportedFOWT("rotor inertial-excitation matrix I_hydro[:3,:3] is not symmetric")
        arm = r3 - np.asarray(fowt.r6, dtype=float)[:3]
        if with_node_arm:
            arm = arm + node_arm(rot.nodeList[0].T)

        def strip(S, a):
            lam, vec = np.linalg.eigh(0.5 * (S + S.T))
            rec = np.zeros(NFIELD)
            rec[F_X:F_X + 3] = r3
            rec[F_AX:F_AX + 3] = a
            rec[F_Q:F_Q + 3], rec[F_P1:F_P1 + 3], rec[F_P2:F_P2 + 3] = vec[:, 0], vec[:, 1], vec[:, 2]
            rec[F_IQ], rec[F_IP1], rec[F_IP2] = lam
            rec[F_MCF] = -1.0
            rec[F_MEM], rec[F_IL] = -1.0, len(rows)
            rows.append(rec)

        strip(I3, arm)
        # [d]x S has no trace for symmetric S, and neither has the block itself: it is a sum of [r]x I terms of the blade
        # members (translateMatrix3to6DOF) rotated as a whole
        if abs(np.trace(M3)) > 1e-9 * max(np.abs(M3).max(), 1e-300):
            raise UnsupportedFOWT("rotor I_hydro[3:,:3] has a trace: not a sum of translated symmetric inertia matrices")
        if np.any(M3):
            coef = _COUPLE_PINV @ M3.reshape(9)
            for kk in range(3):
                S = np.zeros((3, 3))
                for j, (a_, b_) in enumerate(_SYM_IDX):
                    S[a_, b_] = S[b_, a_] = coef[kk * 6 + j]
                if not np.any(S):
                    continue
                e = np.zeros(3)
                e[kk] = 0.5
                strip(S, arm + e)
                strip(-S, arm - e)
    return StripTable(np.array(rows)) if rows else None


def added_mass_morison(strips):
    """A_hydro_morison [6,6] from a strip table -- raft_member.py:1333-1361 and
    raft/helpers.py:537-560 (translateMatrix3to6DOF), in the folded frame.
    Only valid for non-MCF tables when used to cross-check Ca (Cm-1)."""
    A = np.zeros((6, 6))
    for rec in strips:
        q, p1, p2 = rec[F_Q:F_Q + 3], rec[F_P1:F_P1 + 3], rec[F_P2:F_P2 + 3]
        rhoV = rec[F_RHOV]
        ca1 = rec[F_IP1] - rhoV
        ca2 = rec[F_IP2] - rhoV
        Amat = ca1 * np.outer(p1, p1) + ca2 * np.outer(p2, p2) + rec[F_IQ] * np.outer(q, q)
        a = rec[F_AX:F_AX + 3]
        H = np.array([[0, a[2], -a[1]], [-a[2], 0, a[0]], [a[1], -a[0], 0]])
        A[:3, :3] += Amat
        A[:3, 3:] += Amat @ H
        A[3:, :3] += (Amat @ H).T
        A[3:, 3:] += H @ Amat @ H.T
    return A
