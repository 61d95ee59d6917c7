"""TEST INFRASTRUCTURE ONLY -- numpy restatement of the reference's slender-body QTF.

Follows, term by term,
    raft/raft_fowt.py:1988-2078    FOWT.calcQTF_slenderBody   (Pinkster IV term, member loop, Hermitian fill)
    raft/raft_member.py:1488-1674  Member.calcQTF_slenderBody (Rainey / Pinkster strip terms + waterline term)
    raft/helpers.py:239-373        getWaveKin_grad_u1, _grad_dudt, _grad_pres1st, _axdivAcc, _pot2ndOrd
    raft/helpers.py:149-236        getKinematics, getWaveKin
vectorised over the (w1, w2) grid instead of the reference's Python double loop.  Quirks kept on purpose:
  * deg2rad() applied to the (already radian) heading in the gradient / second-order-potential helpers
    (helpers.py:244-245, 289-290, 343-348) but not in the phase of grad_u1 (:259);
  * getWaveKin_axdivAcc removes the axial component of its vel1/vel2 arguments IN PLACE (helpers.py:325-326);
    they are views of nodeV, so every later use of nodeV in the pair loop sees the transverse part only;
  * the waterline term reuses Ca_p1/Ca_p2 of the last submerged strip of the member (raft_member.py:1660-1662);
  * deep-water switch at k h >= 10 in the gradients (helpers.py:252) vs k h > 89.4 in getWaveKin (:215).
The Kim & Yue correction (raft_member.py:1676-1791) is a motion-independent host feeder of the product
(raft_amd.qtf.kay_correction); it is pinned, together with everything here, by the reference's own
*_true_calcQTF_slenderBody.pkl goldens.

Pinning: tests/test_qtf.py checks this file against tests/golden/refgold_qtf_*.npz (the reference pickles,
fixed body, heading 30 deg) and against live-reference QTFs with body motions (tests/golden/qtf_motion_*.npz).
"""
import numpy as np


def _wave_kin(beta, w, k, h, r, rho, g):
    """helpers.py:188-236 with zeta0 = 1: u [3,nw], ud [3,nw], pDyn [nw]."""
    zeta = np.exp(-1j * (k * (np.cos(beta) * r[0] + np.sin(beta) * r[1])))
    z = r[2]
    nw = len(w)
    u = np.zeros((3, nw), dtype=complex)
    pd = np.zeros(nw, dtype=complex)
    if z <= 0:
        Sh, Ch, Cc = np.empty(nw), np.empty(nw), np.empty(nw)
        for i in range(nw):
            if k[i] == 0.0:
                Sh[i], Ch[i], Cc[i] = 1.0, 99999.0, 99999.0
            elif k[i] * h > 89.4:
                Sh[i] = Ch[i] = np.exp(k[i] * z)
                Cc[i] = np.exp(k[i] * z) + np.exp(-k[i] * (z + 2.0 * h))
            else:
                Sh[i] = np.sinh(k[i] * (z + h)) / np.sinh(k[i] * h)
                Ch[i] = np.cosh(k[i] * (z + h)) / np.sinh(k[i] * h)
                Cc[i] = np.cosh(k[i] * (z + h)) / np.cosh(k[i] * h)
        u[0] = w * zeta * Ch * np.cos(beta)
        u[1] = w * zeta * Ch * np.sin(beta)
        u[2] = 1j * w * zeta * Sh
        pd = rho * g * zeta * Cc
    return u, 1j * w * u, pd


def _grad_u1(w, k, beta, h, r):
    """helpers.py:239-277: [3,3,nw]."""
    nw = len(w)
    grad = np.zeros((3, 3, nw), dtype=complex)
    z = r[2]
    cosBeta, sinBeta = np.cos(np.deg2rad(beta)), np.sin(np.deg2rad(beta))        # (sic)
    if z > 0:
        return grad
    ok = k > 0
    kk = np.where(ok, k, 1.0)
    deep = kk * h >= 10
    with np.errstate(over="ignore", invalid="ignore"):
        xy = np.where(deep, np.exp(kk * z), np.cosh(kk * (z + h)) / np.sinh(kk * h))
        zz = np.where(deep, np.exp(kk * z), np.sinh(kk * (z + h)) / np.sinh(kk * h))
    ph = np.exp(-1j * (kk * (np.cos(beta) * r[0] + np.sin(beta) * r[1])))
    aux = w * cosBeta * ph
    grad[0, 0] = -1j * aux * xy * kk * cosBeta
    grad[0, 1] = -1j * aux * xy * kk * sinBeta
    grad[0, 2] = aux * kk * zz
    aux = w * sinBeta * ph
    grad[1, 0] = grad[0, 1]
    grad[1, 1] = -1j * aux * xy * kk * sinBeta
    grad[1, 2] = aux * kk * zz
    aux = 1j * w * ph
    grad[2, 0] = grad[0, 2]
    grad[2, 1] = grad[0, 1]                                                       # (sic) helpers.py:274
    grad[2, 2] = aux * kk * xy
    grad[:, :, ~ok] = 0
    return grad


def _grad_pres1st(k, beta, h, r, rho, g):
    """helpers.py:283-308: [3,nw]."""
    nw = len(k)
    grad = np.zeros((3, nw), dtype=complex)
    z = r[2]
    cosBeta, sinBeta = np.cos(np.deg2rad(beta)), np.sin(np.deg2rad(beta))        # (sic), also in the phase
    if z > 0:
        return grad
    ok = k > 0
    kk = np.where(ok, k, 1.0)
    deep = kk * h >= 10
    with np.errstate(over="ignore", invalid="ignore"):
        xy = np.where(deep, np.exp(kk * z), np.cosh(kk * (z + h)) / np.cosh(kk * h))
        zz = np.where(deep, np.exp(kk * z), np.sinh(kk * (z + h)) / np.cosh(kk * h))
    ph = np.exp(-1j * (kk * (cosBeta * r[0] + sinBeta * r[1])))
    grad[0] = rho * g * xy * ph * (-1j * kk * cosBeta)
    grad[1] = rho * g * xy * ph * (-1j * kk * sinBeta)
    grad[2] = rho * g * zz * ph * kk
    grad[:, ~ok] = 0
    return grad


def _pot2nd(w, k, beta, h, r, g, rho):
    """helpers.py:337-373 on the full grid: acc [3,nw,nw], p [nw,nw] (index order [i1,i2])."""
    nw = len(w)
    acc = np.zeros((3, nw, nw), dtype=complex)
    p = np.zeros((nw, nw), dtype=complex)
    z = r[2]
    if z > 0:
        return acc, p
    w1, w2 = w[:, None], w[None, :]
    k1, k2 = k[:, None], k[None, :]
    cosB, sinB = np.cos(np.deg2rad(beta)), np.sin(np.deg2rad(beta))               # (sic)
    ok = (w1 != w2) & (k1 > 0) & (k2 > 0)
    kx, ky = (k1 - k2) * cosB, (k1 - k2) * sinB
    nrm = np.sqrt(kx ** 2 + ky ** 2)
    with np.errstate(divide="ignore", invalid="ignore", over="ignore"):
        th1, th2 = np.tanh(k1 * h), np.tanh(k2 * h)
        den = (w1 - w2) ** 2 / g - nrm * np.tanh(nrm * h)
        g12 = (-1j * g / (2 * w1)) * ((k1 ** 2) * (1 - th1 ** 2) - 2 * k1 * k2 * (1 + th1 * th2)) / den
        g21 = (-1j * g / (2 * w2)) * ((k2 ** 2) * (1 - th2 ** 2) - 2 * k2 * k1 * (1 + th2 * th1)) / den
        aux = 0.5 * (g21 + np.conj(g12))
        xy = np.cosh(nrm * (z + h)) / np.cosh(nrm * h)
        zz = np.sinh(nrm * (z + h)) / np.cosh(nrm * h)
        ph = np.exp(-1j * (kx * r[0] + ky * r[1]))
        a0 = aux * xy * ph * (w1 - w2) * kx
        a1 = aux * xy * ph * (w1 - w2) * ky
        a2 = aux * zz * ph * 1j * (w1 - w2) * nrm
        pp = aux * xy * ph * (-1j) * rho * (w1 - w2)
    acc[0], acc[1], acc[2] = np.where(ok, a0, 0), np.where(ok, a1, 0), np.where(ok, a2, 0)
    p = np.where(ok, pp, 0)
    return acc, p


def _cross_mat(v):
    """-getH(v) (helpers.py:428-437): the matrix of v x . ; v [3,nw] -> [3,3,nw]."""
    Z = np.zeros_like(v[0])
    return np.array([[Z, -v[2], v[1]], [v[2], Z, -v[0]], [-v[1], v[0], Z]])


def _mv(M, v):
    """[3,3,...] @ [3,...] with numpy broadcasting on the trailing axes."""
    return np.array([M[i, 0] * v[0] + M[i, 1] * v[1] + M[i, 2] * v[2] for i in range(3)])


def _cmv(M, v):
    """constant [3,3] @ [3,...]"""
    return np.tensordot(M, v, axes=(1, 0))


def _to6(f3, r, F):
    """translateForce3to6DOF (helpers.py:468-483) accumulated into F [6,nw,nw]."""
    F[0:3] += f3
    F[3] += r[1] * f3[2] - r[2] * f3[1]
    F[4] += r[2] * f3[0] - r[0] * f3[2]
    F[5] += r[0] * f3[1] - r[1] * f3[0]


def qtf_slender_body(tab, Xi, beta, w, k, h, rho, g, M_struc, kay=None):
    """FOWT.calcQTF_slenderBody for one heading: qtf [nw,nw,6] (Hermitian-completed).
    tab: raft_amd.qtf.QtfTable; Xi [6,nw] motion RAOs on the 2nd-order grid (zeros = fixed body);
    kay: optional [nw,nw,6] Kim & Yue table (upper triangle)."""
    nw = len(w)
    Xi = np.asarray(Xi, dtype=complex)
    Q = np.zeros((6, nw, nw), dtype=complex)
    up = (w[None, :] >= w[:, None])                                # i2 >= i1 (raft_member.py:1543-1544)
    one = lambda a: a[..., :, None]                                # index i1
    two = lambda a: a[..., None, :]                                # index i2

    # Pinkster IV: rotation of the first-order inertial forces (raft_fowt.py:2044-2062)
    F1st = np.matmul(np.asarray(M_struc, dtype=float), (-w ** 2 * Xi))
    th = Xi[3:]
    for blk in (slice(0, 3), slice(3, 6)):
        a = np.cross(one(th), np.conj(two(F1st[blk])), axis=0)
        b = np.cross(np.conj(two(th)), one(F1st[blk]), axis=0)
        Q[blk] += 0.25 * (a + b)

    OM = _cross_mat(1j * w * th)                                   # OMEGA = -getH(i w theta)  (:1588-1589)
    for rec in tab.strips:
        r, q, p1, p2 = rec[0:3], rec[3:6], rec[6:9], rec[9:12]
        Ca1, Ca2, CaE, v_i, v_e, a_i = rec[12:18]
        p1M, p2M, qM = np.outer(p1, p1), np.outer(p2, p2), np.outer(q, q)
        P1 = (1.0 + Ca1) * p1M + (1.0 + Ca2) * p2M
        Pa = Ca1 * p1M + Ca2 * p2M
        # per-frequency kinematics at the strip (raft_member.py:1509-1519)
        dr = Xi[:3] + np.cross(th, r, axis=0)                      # SmallRotate(r, th) = th x r
        nodeV = 1j * w * dr
        u, _, _ = _wave_kin(beta, w, k, h, r, rho, g)
        gu = _grad_u1(w, k, beta, h, r)
        gdudt = 1j * w * gu
        nar = np.tensordot(q, u - nodeV, axes=(0, 0))              # nodeV_axial_rel
        gp = _grad_pres1st(k, beta, h, r, rho, g)
        nodeVt = nodeV - np.tensordot(q, nodeV, axes=(0, 0)) * q[:, None]     # in-place quirk of axdivAcc

        F2 = np.zeros((6, nw, nw), dtype=complex)
        acc2, p2nd = _pot2nd(w, k, beta, h, r, g, rho)
        f_2nd = rho * v_i * _cmv(P1, acc2)
        conv = 0.25 * (_mv(one(gu), np.conj(two(u))) + _mv(np.conj(two(gu)), one(u)))
        f_conv = rho * v_i * _cmv(P1, conv)
        # axial-divergence acceleration (helpers.py:311-335)
        dwdz = np.tensordot(q, np.tensordot(gu, q, axes=(1, 0)), axes=(0, 0))     # q . (grad_u q)
        ut = u - np.tensordot(q, u, axes=(0, 0)) * q[:, None]
        ax = 0.25 * (one(dwdz) * np.conj(two(ut - nodeVt)) + np.conj(two(dwdz)) * one(ut - nodeVt))
        ax = ax - np.tensordot(q, ax, axes=(0, 0)) * q[:, None, None]
        f_axdv = rho * v_i * _cmv(Pa, ax)
        nab = 0.25 * (_mv(one(gdudt), np.conj(two(dr))) + _mv(np.conj(two(gdudt)), one(dr)))
        f_nab = rho * v_i * _cmv(P1, nab)
        qv = q[:, None, None]
        rs = _mv(one(OM), np.conj(two(nar) * qv)) + _mv(np.conj(two(OM)), one(nar) * qv)
        f_rslb = -0.25 * 2 * _cmv(Pa, rs) * rho * v_i
        u1a, u2a = one(u - nodeVt), two(u - nodeVt)
        V1, V2 = one(gu + OM), two(gu + OM)
        aux = 0.25 * (_mv(V1, np.conj(_cmv(Pa, u2a))) + _mv(np.conj(V2), _cmv(Pa, u1a)))
        aux = aux - _cmv(qM, aux)
        f_rslb = f_rslb + rho * v_i * aux
        u1t, u2t = u1a - _cmv(qM, u1a), u2a - _cmv(qM, u2a)
        aux = 0.25 * (_cmv(Pa, _mv(V1, np.conj(u2t))) + _cmv(Pa, _mv(np.conj(V2), u1t)))
        f_rslb = f_rslb - rho * v_i * aux
        # end effects (raft_member.py:1613-1627)
        f_2nd = f_2nd + a_i * p2nd * qv + rho * v_e * CaE * _cmv(qM, acc2)
        f_conv = f_conv + rho * v_e * CaE * _cmv(qM, conv)
        f_nab = f_nab + rho * v_e * CaE * _cmv(qM, nab)
        p_nab = 0.25 * (np.sum(one(gp) * np.conj(two(dr)), axis=0) + np.sum(np.conj(two(gp)) * one(dr), axis=0))
        f_nab = f_nab + a_i * p_nab * qv
        p_drop = -2 * 0.25 * 0.5 * rho * np.sum(_cmv(p1M + p2M, u1a) * np.conj(_cmv(Pa, u2a)), axis=0)
        f_conv = f_conv + a_i * p_drop * qv
        u1p, u2p = _cmv(Pa, u1t), _cmv(Pa, u2t)
        f_conv = f_conv + 0.25 * a_i * rho * (np.conj(u1p) * two(nar) + u2p * np.conj(one(nar)))
        for f3 in (f_2nd, f_conv, f_axdv, f_nab, f_rslb):
            _to6(f3, r, F2)
        Q += F2

    # waterline term of every member that crosses z = 0 (raft_member.py:1517-1534, 1635-1668)
    for m in tab.members:
        if m[0] == 0.0:
            continue
        r_int, a_wl, Ca1, Ca2, p1, p2 = m[1:4], m[4], m[5], m[6], m[7:10], m[10:13]
        p1M, p2M = np.outer(p1, p1), np.outer(p2, p2)
        _, ud_wl, eta = _wave_kin(beta, w, k, h, r_int, 1.0, 1.0)
        dr_wl = Xi[:3] + np.cross(th, r_int, axis=0)
        a_b = 1j * w * (1j * w * dr_wl)
        g_e1 = -g * (np.cross(th, p1, axis=0)[2] * p1[:, None] + np.cross(th, p2, axis=0)[2] * p2[:, None])
        eta_r = eta - dr_wl[2]
        f = 0.25 * (one(ud_wl) * np.conj(two(eta_r)) + np.conj(two(ud_wl)) * one(eta_r))
        f = rho * a_wl * _cmv((1.0 + Ca1) * p1M + (1.0 + Ca2) * p2M, f)
        a_eta = 0.25 * (one(a_b) * np.conj(two(eta_r)) + np.conj(two(a_b)) * one(eta_r))
        f = f - rho * a_wl * _cmv(Ca1 * p1M + Ca2 * p2M, a_eta)
        f = f - 0.25 * rho * a_wl * (one(g_e1) * np.conj(two(eta_r)) + np.conj(two(g_e1)) * one(eta_r))
        F = np.zeros((6, nw, nw), dtype=complex)
        _to6(f, r_int, F)
        Q += F

    Q = np.where(up[None], Q, 0.0)
    qtf = np.transpose(Q, (1, 2, 0)).copy()
    if kay is not None:
        qtf += kay
    for i in range(6):                                             # Hermitian fill (raft_fowt.py:2069-2070)
        qi = qtf[:, :, i]
        qtf[:, :, i] = qi + np.conj(qi).T - np.diag(np.diag(np.conj(qi)))
    return qtf
