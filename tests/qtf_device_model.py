"""TEST INFRASTRUCTURE -- numpy model of the FORMULATION k_qtf_pairs evaluates (raft_amd/csrc/raftx_qtf.h): the strip
terms of Member.calcQTF_slenderBody (raft/raft_member.py:1541-1633) rewritten in each strip's own frame (p1, p2, q).

Per (strip, frequency) the table kernel leaves, in that frame: u, dr (3), ua = (u - nodeV_t) (components 1, 2: its axial
component never enters), grad_pres (3), G = E^T grad_u E (9), S = E^T [i w theta]x E (9; M = G + S), nodeV_axial_rel, dwdz,
and the phase factor of the second-order potential.  Projections become component selections (P_Ca v = (Ca1 v1, Ca2 v2, 0),
removing the axial part = dropping component 3), so a strip-pair is 50 complex multiply-adds and three complex scalars
(f1, f2, f3) that meet the strip's constant 6-vectors [e_i ; r x e_i] once.

tests/test_qtf.py checks this model against the term-by-term restatement of the reference (oracle/qtf_oracle.py); the
device kernel is checked against both on the GPU (tests/test_hip_qtf.py)."""
import numpy as np

from oracle import qtf_oracle as qo


def strip_tables(rec, Xi, beta, w, k, h, rho, g):
    """Member-frame first-order quantities of one strip, every one [.., nw]."""
    r, q, p1, p2 = rec[0:3], rec[3:6], rec[6:9], rec[9:12]
    E = np.stack([p1, p2, q], axis=1)                               # columns e_1, e_2, e_3
    th = Xi[3:]
    dr = Xi[:3] + np.cross(th, r, axis=0)
    nodeV = 1j * w * dr
    u, _, _ = qo._wave_kin(beta, w, k, h, r, rho, g)
    gu = qo._grad_u1(w, k, beta, h, r)
    gp = qo._grad_pres1st(k, beta, h, r, rho, g)
    nar = np.tensordot(q, u - nodeV, axes=(0, 0))
    nodeVt = nodeV - np.tensordot(q, nodeV, axes=(0, 0)) * q[:, None]
    dz = np.tensordot(q, np.tensordot(gu, q, axes=(1, 0)), axes=(0, 0))
    OM = qo._cross_mat(1j * w * th)
    cB, sB = np.cos(np.deg2rad(beta)), np.sin(np.deg2rad(beta))
    e2 = np.exp(-1j * (k * (cB * r[0] + sB * r[1])))
    fr = lambda v: np.tensordot(E.T, v, axes=(1, 0))                 # E^T v
    frm = lambda M: np.einsum("ai,abw,bj->ijw", E, M, E)             # E^T M E
    return dict(u=fr(u), dr=fr(dr), x=fr(u - nodeVt)[:2], gp=fr(gp), nar=nar, dz=dz, e2=e2, G=frm(gu), S=frm(OM),
                hx=cB * E[0] + sB * E[1], hz=E[2], W=np.stack([np.concatenate([E[:, i], np.cross(r, E[:, i])]) for i in range(3)]))


def qtf_strips(tab, Xi, beta, w, k, h, rho, g):
    """Sum over the strips of the strip terms, [6, nw, nw] on the upper triangle (i2 >= i1), zeros below."""
    nw = len(w)
    Xi = np.asarray(Xi, dtype=complex)
    one = lambda a: a[..., :, None]
    two = lambda a: a[..., None, :]
    w1, w2, k1, k2 = w[:, None], w[None, :], k[:, None], k[None, :]
    cB, sB = np.cos(np.deg2rad(beta)), np.sin(np.deg2rad(beta))
    pot = (w1 != w2) & (k1 > 0) & (k2 > 0)
    kx, ky = (k1 - k2) * cB, (k1 - k2) * sB
    nrm = np.sqrt(kx ** 2 + ky ** 2)
    with np.errstate(divide="ignore", invalid="ignore", over="ignore"):
        t1, t2 = np.tanh(k1 * h), np.tanh(k2 * h)
        den = (w1 - w2) ** 2 / g - nrm * np.tanh(nrm * h)
        g12 = (-g / (2 * w1)) * ((k1 ** 2) * (1 - t1 ** 2) - 2 * k1 * k2 * (1 + t1 * t2)) / den
        g21 = (-g / (2 * w2)) * ((k2 ** 2) * (1 - t2 ** 2) - 2 * k2 * k1 * (1 + t2 * t1)) / den
    paux = np.where(pot, 0.5j * (g21 - g12), 0.0)
    dwp = w1 - w2
    Q = np.zeros((6, nw, nw), dtype=complex)
    for rec in tab.strips:
        T = strip_tables(rec, Xi, beta, w, k, h, rho, g)
        z = rec[2]
        Ca = rec[12:14]
        CaE, v_i, v_e, a_i = rec[14:18]
        # second-order potential (pair scalars; cosh / sinh of nrm (z + h) depend on the strip's depth only)
        with np.errstate(over="ignore", invalid="ignore"):
            xy = np.cosh(nrm * (z + h)) / np.cosh(nrm * h)
            zz = np.sinh(nrm * (z + h)) / np.cosh(nrm * h)
        base = paux * one(T["e2"]) * np.conj(two(T["e2"])) if z <= 0 else np.zeros((nw, nw), dtype=complex)
        base = np.where(pot, base, 0.0)
        acc2 = [base * dwp * (xy * (k1 - k2) * T["hx"][i] + 1j * zz * nrm * T["hz"][i]) for i in range(3)]
        acc2 = [np.where(pot, a, 0.0) for a in acc2]
        p2nd = np.where(pot, -1j * base * xy * rho * dwp, 0.0)
        # convective + body-motion-in-the-wave-field accelerations through ONE product per side
        a = np.conj(two(T["u"])) + 1j * w1 * np.conj(two(T["dr"]))
        b = one(T["u"]) - 1j * w2 * one(T["dr"])
        G1, G2c = one(T["G"]), np.conj(two(T["G"]))
        A = [acc2[i] + 0.25 * sum(G1[i, j] * a[j] + G2c[i, j] * b[j] for j in range(3)) for i in range(3)]
        x1, x2 = one(T["x"]), two(T["x"])
        x2c = np.conj(x2)
        M1, M2c = one(T["G"] + T["S"]), np.conj(two(T["G"] + T["S"]))
        S1, S2c = one(T["S"]), np.conj(two(T["S"]))
        nar1, nar2c = one(T["nar"]), np.conj(two(T["nar"]))
        f = []
        for i in range(2):
            t = [M1[i, j] * x2c[j] + M2c[i, j] * x1[j] for j in range(2)]
            aux = 0.25 * (Ca[0] * t[0] + Ca[1] * t[1])
            aux2 = 0.25 * Ca[i] * (t[0] + t[1])
            ax0 = 0.25 * (x2c[i] * one(T["dz"]) + x1[i] * np.conj(two(T["dz"])))
            rs = S1[i, 2] * nar2c + S2c[i, 2] * nar1
            fi = rho * v_i * ((1.0 + Ca[i]) * A[i] + Ca[i] * ax0 - 0.5 * Ca[i] * rs + aux - aux2)
            fi = fi + 0.25 * a_i * rho * Ca[i] * (np.conj(x1[i]) * np.conj(nar2c) + x2[i] * np.conj(nar1))
            f.append(fi)
        p_nab = 0.25 * sum(one(T["gp"])[j] * np.conj(two(T["dr"]))[j] + np.conj(two(T["gp"]))[j] * one(T["dr"])[j] for j in range(3))
        p_drop = -0.25 * rho * sum(x1[i] * Ca[i] * x2c[i] for i in range(2))
        f.append(rho * v_e * CaE * A[2] + a_i * (p2nd + p_nab + p_drop))
        for i in range(3):
            Q += T["W"][i][:, None, None] * f[i][None]
    up = (w[None, :] >= w[:, None])
    return np.where(up[None], Q, 0.0)
