"""Second-order (difference-frequency) slender-body QTF: host side.

Packs what the QTF kernels consume out of a FOWT's members, evaluates the host-only feeders
(Kim & Yue analytical correction with SciPy Hankel functions), and mirrors

    raft/raft_fowt.py:1988-2078    FOWT.calcQTF_slenderBody(waveHeadInd, Xi0=None, ...)
    raft/raft_member.py:1488-1674  Member.calcQTF_slenderBody
    raft/raft_member.py:1676-1791  Member.correction_KAY
    raft/raft_fowt.py:2158-2253    FOWT.calcHydroForce_2ndOrd

Record layouts (doubles):
  strip  (QS_N = 24): 0-2 r (absolute; also the moment arm, raft_member.py:1629-1633), 3-5 q, 6-8 p1, 9-11 p2,
                      12 Ca_p1, 13 Ca_p2, 14 Ca_End (interpolated at the strip, :1557-1559),
                      15 v_side (with the waterline scaling of :1562-1568), 16 v_end (:1613-1618),
                      17 a_i (Member.a_i[il], raft_member.py:1343,1347), 18 member index
  member (QM_N = 16): 0 crosses-waterline flag (:1523,1638), 1-3 r_int (:1524), 4 waterline area (:1641-1657),
                      5-6 Ca_p1, Ca_p2 used by the waterline term -- the reference reuses the values left over
                      from the LAST submerged strip of the member (:1660-1662), 7-9 p1, 10-12 p2
Only strips with z < 0 are packed (:1553).
"""
import numpy as np

QS_N = 24
QM_N = 16


class QtfTable:
    def __init__(self, strips, members, kay_geom):
        self.strips = np.ascontiguousarray(strips, dtype=np.float64).reshape(-1, QS_N)
        self.members = np.ascontiguousarray(members, dtype=np.float64).reshape(-1, QM_N)
        self.kay_geom = kay_geom          # per MCF member: dict(rA, rB, r, ds, dls, p1, p2)


def _ends(mem):
    rA = np.asarray(getattr(mem, "rA", mem.r[0]), dtype=float)
    rB = np.asarray(getattr(mem, "rB", mem.r[-1]), dtype=float)
    return rA, rB


def pack_qtf(fowt, memberList=None):
    members = fowt.memberList if memberList is None else memberList
    srec, mrec, kay = [], [], []
    for im, mem in enumerate(members):
        rA, rB = _ends(mem)
        if rA[2] > 0 and rB[2] > 0:                       # raft_member.py:1493-1494
            continue
        circ = mem.shape == "circular"
        r = np.asarray(mem.r, dtype=float)
        q, p1, p2 = (np.asarray(v, dtype=float) for v in (mem.q, mem.p1, mem.p2))
        midx = len(mrec)
        last_ca = (0.0, 0.0)
        for il in range(mem.ns):
            if r[il, 2] >= 0:
                continue
            Ca_p1 = float(np.interp(mem.ls[il], mem.stations, mem.Ca_p1))
            Ca_p2 = float(np.interp(mem.ls[il], mem.stations, mem.Ca_p2))
            Ca_End = float(np.interp(mem.ls[il], mem.stations, mem.Ca_End))
            last_ca = (Ca_p1, Ca_p2)
            d, dr, dl = mem.ds[il], mem.drs[il], float(mem.dls[il])
            if circ:
                v_i = 0.25 * np.pi * d ** 2 * dl
            else:
                v_i = d[0] * d[1] * dl
            if r[il, 2] + 0.5 * dl > 0:
                v_i = v_i * (0.5 * dl - r[il, 2]) / dl
            if circ:
                v_e = np.pi / 12.0 * abs((d + dr) ** 3 - (d - dr) ** 3)
                a_i = np.pi * d * dr
            else:
                v_e = np.pi / 12.0 * ((np.mean(d + dr)) ** 3 - (np.mean(d - dr)) ** 3)
                a_i = (d[0] + dr[0]) * (d[1] + dr[1]) - (d[0] - dr[0]) * (d[1] - dr[1])
            rec = np.zeros(QS_N)
            rec[0:3], rec[3:6], rec[6:9], rec[9:12] = r[il], q, p1, p2
            rec[12:19] = [Ca_p1, Ca_p2, Ca_End, float(v_i), float(v_e), float(a_i), midx]
            srec.append(rec)
        m = np.zeros(QM_N)
        if r[-1, 2] * r[0, 2] < 0:                         # raft_member.py:1523
            m[0] = 1.0
            m[1:4] = r[0] + (r[-1] - r[0]) * (0.0 - r[0, 2]) / (r[-1, 2] - r[0, 2])
            i_wl = np.where(r[:, 2] < 0)[0][-1]
            ds = np.asarray(mem.ds, dtype=float)
            if circ:
                d_wl = 0.5 * (ds[i_wl] + ds[i_wl + 1]) if i_wl != len(ds) - 1 else ds[i_wl]
                m[4] = 0.25 * np.pi * d_wl ** 2
            else:
                if i_wl != len(ds) - 1:
                    d1, d2 = 0.5 * (ds[i_wl, 0] + ds[i_wl + 1, 0]), 0.5 * (ds[i_wl, 1] + ds[i_wl + 1, 1])
                else:
                    d1, d2 = ds[i_wl, 0], ds[i_wl, 1]
                m[4] = d1 * d2
            m[5], m[6] = last_ca
        m[7:10], m[10:13] = p1, p2
        mrec.append(m)
        if bool(getattr(mem, "MCF", False)):
            kay.append(dict(rA=rA, rB=rB, r=r, ds=np.asarray(mem.ds, dtype=float), dls=np.asarray(mem.dls, dtype=float),
                            p1=p1, p2=p2))
    strips = np.array(srec) if srec else np.zeros((0, QS_N))
    return QtfTable(strips, np.array(mrec) if mrec else np.zeros((0, QM_N)), kay)


# ---------------------------------------------------------------------- Kim & Yue correction (host feeder)
QK_N = 12          # doubles per Kim & Yue item (include/raftx.h RAFTX_QK_*)


def kay_items(kay_geom, beta):
    """Geometry of Member.correction_KAY (raft_member.py:1676-1791) as flat records for raftx_qtf_kay: for every
    MacCamy-Fuchs member that crosses the waterline one WATERLINE item and one item per submerged segment between
    consecutive strip nodes.  Record: R, kind (0 waterline / 1 segment), z1, z2, moment arm (3), unit force direction
    pforce (3, depends on the heading), waterline point x, y (the phase reference of ALL items of the member, :1783)."""
    cosB, sinB = np.cos(beta), np.sin(beta)
    rows = []
    for gm in kay_geom:
        rA, rB, r, ds, dls, p1, p2 = (gm[x] for x in ("rA", "rB", "r", "ds", "dls", "p1", "p2"))
        if rA[2] * rB[2] >= 0:
            continue
        bvec = np.array([cosB, sinB, 0.0])
        pforce = np.dot(bvec, p1) * p1 + np.dot(bvec, p2) * p2
        pforce = pforce / np.linalg.norm(pforce)
        rwl = rA + (rB - rA) * (0 - rA[2]) / (rB[2] - rA[2])
        R = np.interp(0, r[:, 2], 0.5 * ds)
        rows.append([R, 0.0, 0.0, 0.0, rwl[0], rwl[1], rwl[2], pforce[0], pforce[1], pforce[2], rwl[0], rwl[1]])
        for il in range(len(r) - 1):
            z1 = r[il, 2]
            if z1 > 0:
                continue
            z2 = min(r[il + 1, 2], 0.0)
            R1 = ds[il] / 2 if dls[il] != 0 else ds[il]
            R2 = ds[il + 1] / 2 if dls[il + 1] != 0 else ds[il]
            mid = 0.5 * (r[il] + r[il + 1])
            rows.append([0.5 * (R1 + R2), 1.0, z1, z2, mid[0], mid[1], mid[2], pforce[0], pforce[1], pforce[2], rwl[0], rwl[1]])
    return np.array(rows, dtype=np.float64).reshape(-1, QK_N)


def kay_correction(kay_geom, w, k, beta, h, rho=1025.0, g=9.81, Nm=10):
    """Sum over the MCF members of Member.correction_KAY (raft_member.py:1676-1791) for every pair i2 >= i1:
    [nw,nw,6] complex, zero below the diagonal.  Independent of the body motions -> a per-design table."""
    from scipy.special import hankel1
    nw = len(w)
    out = np.zeros((nw, nw, 6), dtype=complex)
    if not kay_geom:
        return out
    w1, w2 = w[:, None], w[None, :]
    k1, k2 = k[:, None], k[None, :]
    up = w2 >= w1

    def dhankel(n, x):
        """H_n'(x) of the Hankel function of the first kind (recurrence H' = (H_{n-1} - H_{n+1}) / 2)"""
        return 0.5 * (hankel1(n - 1, x) - hankel1(n + 1, x))

    def omega(x1, x2, n):
        # the n-th term of the Kim & Yue sum for the pair (k1 R, k2 R): 1/(H'_{n+1}(x1) conj H'_n(x2)) - 1/(H'_n(x1) conj H'_{n+1}(x2))
        # (raft_member.py:1688-1695); one table of derivatives per argument
        d1, d1p = dhankel(n, x1), dhankel(n + 1, x1)
        d2, d2p = np.conj(dhankel(n, x2)), np.conj(dhankel(n + 1, x2))
        return 1.0 / (d1p * d2) - 1.0 / (d1 * d2p)

    cosB, sinB = np.cos(beta), np.sin(beta)
    kd = k1 - k2                                           # k1_k2 = (k1-k2) (cosB, sinB, 0)
    for gm in kay_geom:
        rA, rB, r, ds, dls, p1, p2 = (gm[x] for x in ("rA", "rB", "r", "ds", "dls", "p1", "p2"))
        F = np.zeros((nw, nw, 6), dtype=complex)
        if rA[2] * rB[2] >= 0:
            continue
        bvec = np.array([cosB, sinB, 0.0])
        pforce = np.dot(bvec, p1) * p1 + np.dot(bvec, p2) * p2
        pforce = pforce / np.linalg.norm(pforce)
        rwl = rA + (rB - rA) * (0 - rA[2]) / (rB[2] - rA[2])
        R = np.interp(0, r[:, 2], 0.5 * ds)
        k1R, k2R = k1 * R, k2 * R
        Fwl = np.zeros((nw, nw), dtype=complex)
        for nn in range(Nm + 1):
            Fwl = Fwl + (-rho * g * R * 2j / np.pi / (k1R * k2R) * omega(k1R, k2R, nn))
        phase_wl = np.exp(-1j * kd * (cosB * rwl[0] + sinB * rwl[1]))
        Fwl = np.real(Fwl) * phase_wl

        def add(Fs, r0):
            f3 = Fs[:, :, None] * pforce[None, None, :]
            F[:, :, :3] += f3
            F[:, :, 3:] += np.cross(np.broadcast_to(r0, f3.shape), f3)

        add(Fwl, rwl)
        for il in range(len(r) - 1):
            z1 = r[il, 2]
            if z1 > 0:
                continue
            z2 = min(r[il + 1, 2], 0.0)
            R1 = ds[il] / 2 if dls[il] != 0 else ds[il]
            R2 = ds[il + 1] / 2 if dls[il + 1] != 0 else ds[il]
            R = 0.5 * (R1 + R2)
            k1R, k2R = k1 * R, k2 * R
            H = h / R
            k1h, k2h = k1R * H, k2R * H
            ks, kdh = k1 + k2, k1h - k2h
            sp2, sp1 = np.sinh(ks * (z2 + h)) / (k1h + k2h), np.sinh(ks * (z1 + h)) / (k1h + k2h)
            with np.errstate(divide="ignore", invalid="ignore"):
                sm2 = np.where(w1 == w2, (z2 + h) / h, np.sinh((k1 - k2) * (z2 + h)) / kdh)
                sm1 = np.where(w1 == w2, (z1 + h) / h, np.sinh((k1 - k2) * (z1 + h)) / kdh)
            Im = 0.5 * (sp2 - sm2 - sp1 + sm1)
            Ip = 0.5 * (sp2 + sm2 - sp1 - sm1)
            c1, c2 = np.cosh(k1h), np.cosh(k2h)
            dF = np.zeros((nw, nw), dtype=complex)
            for nn in range(Nm + 1):
                dF = dF + rho * g * R * 2j / np.pi / (k1R * k2R) * omega(k1R, k2R, nn) * (
                    k1h * k2h / np.sqrt(k1h * np.tanh(k1h)) / np.sqrt(k2h * np.tanh(k2h))
                    * (Im + Ip * nn * (nn + 1) / k1R / k2R) / c1 / c2)
            dF = np.real(dF) * phase_wl                     # (sic) phase of the waterline point, :1783
            add(dF, 0.5 * (r[il] + r[il + 1]))
        F = np.where((k1 < k2)[:, :, None], np.conj(F), F)  # :1787-1788
        out += np.where(up[:, :, None], F, 0.0)
    return out


# ---------------------------------------------------------------------- second-order force from the QTF
def hydro_force_2nd(qtf, w2nd, w, dw, S0):
    """FOWT.calcHydroForce_2ndOrd, interpMode='qtf', one heading (raft_fowt.py:2173-2178, 2209-2245):
    qtf [n2,n2,6] on the grid w2nd -> (f_mean [6], f [6,nw]) on the first-order grid w."""
    from scipy.interpolate import RegularGridInterpolator
    nw = len(w)
    f = np.zeros([6, nw])
    f_mean = np.zeros(6)
    w_mesh = np.meshgrid(w, w, indexing="ij")
    points = np.array([w_mesh[0].ravel(), w_mesh[1].ravel()]).T
    for idof in range(6):
        re = RegularGridInterpolator((w2nd, w2nd), qtf[:, :, idof].real, bounds_error=False, fill_value=0)(points)
        im = RegularGridInterpolator((w2nd, w2nd), qtf[:, :, idof].imag, bounds_error=False, fill_value=0)(points)
        q = (re + 1j * im).reshape(nw, nw)
        for imu in range(1, nw):
            Saux = np.zeros(nw)
            Saux[0:nw - imu] = S0[imu:]
            Qaux = np.zeros(nw, dtype=complex)
            Qaux[0:nw - imu] = np.diag(q, imu)
            f[idof, imu] = 4 * np.sqrt(np.sum(S0 * Saux * np.abs(Qaux) ** 2)) * dw
        f_mean[idof] = 2 * np.sum(S0 * np.diag(q.real, 0)) * dw
    f[:, 0:-1] = f[:, 1:]
    f[:, -1] = 0
    return f_mean, f


def hydro_force_2nd_spectrum(qtf, w2nd, w, dw, S0):
    """interpMode='spectrum' branch of FOWT.calcHydroForce_2ndOrd (raft_fowt.py:2186-2207)."""
    nw, nw1 = len(w), len(w2nd)
    S = np.interp(w2nd, w, S0, left=0, right=0)
    mu = w2nd - w2nd[0]
    Sf = np.zeros([6, nw1])
    f = np.zeros([6, nw], dtype=complex)
    f_mean = np.zeros(6)
    for idof in range(6):
        for imu in range(1, nw1):
            Saux = np.zeros(nw1)
            Saux[0:nw1 - imu] = S[imu:]
            Qaux = np.zeros(nw1, dtype=complex)
            Qaux[0:nw1 - imu] = np.diag(qtf[:, :, idof], imu)
            Sf[idof, imu] = 8 * np.sum(S * Saux * np.abs(Qaux) ** 2) * (w2nd[1] - w2nd[0])
        f_mean[idof] = 2 * np.sum(S * np.diag(qtf[:, :, idof].real, 0)) * (w2nd[1] - w2nd[0])
        f[idof, :] = np.sqrt(2 * np.interp(w - w[0], mu, Sf[idof, :], left=0, right=0) * dw)
    f[:, 0:-1] = f[:, 1:]
    f[:, -1] = 0
    return f_mean, f


def interp_heading(qtf4, heads, beta):
    """Heading interpolation of raft_fowt.py:2173-2178: qtf4 [n,n,nHeads,6] -> [n,n,6]."""
    if len(heads) == 1:
        return qtf4[:, :, 0, :]
    from scipy.interpolate import interp1d
    re = interp1d(heads, qtf4.real, assume_sorted=True, axis=2, bounds_error=False,
                  fill_value=(qtf4[:, :, 0, :].real, qtf4[:, :, -1, :].real))(beta)
    im = interp1d(heads, qtf4.imag, assume_sorted=True, axis=2, bounds_error=False,
                  fill_value=(qtf4[:, :, 0, :].imag, qtf4[:, :, -1, :].imag))(beta)
    return re + 1j * im


# ---------------------------------------------------------------------- WAMIT-format files next to the path
def write_qtf12d(path, qtf4, w, heads, rho, g):
    """FOWT.writeQTF (raft_fowt.py:2131-2155): WAMIT .12d, upper triangle, periods, ULEN = 1."""
    with open(path, "w") as f:
        for ih in range(len(heads)):
            for iDoF in range(qtf4.shape[3]):
                q = qtf4[:, :, ih, iDoF]
                for i1 in range(len(w)):
                    for i2 in range(i1, len(w)):
                        F = q[i1, i2] / (rho * g)
                        f.write(f"{2*np.pi/w[i1]: 8.4e} {2*np.pi/w[i2]: 8.4e} {np.rad2deg(heads[ih]): 8.4e} {np.rad2deg(heads[ih]): 8.4e} "
                                f"{iDoF+1} {np.abs(F): 8.4e} {np.angle(F): 8.4e} {F.real: 8.4e} {F.imag: 8.4e}\n")


def read_qtf12d(path, rho, g, nDOF=6, ULEN=1):
    """WAMIT ``.12d`` difference-frequency QTF file -> (heads_2nd [rad], w_2nd [rad/s], qtf [n,n,nHeads,nDOF]), as
    FOWT.readQTF stores them (raft_fowt.py:2081-2128; same values, checked bit for bit against it on the reference's
    ``marin_semi.12d``).  Columns: PER1 PER2 BETA1 BETA2 I MOD PHA RE IM, one triangle of the Hermitian matrix; forces
    are dimensionalised with rho g ULEN, moments (I >= 4) with rho g ULEN^2.  The whole table is placed with one
    fancy-indexed assignment per triangle (rows in file order, so a repeated entry keeps its last occurrence, as a
    row-by-row fill would)."""
    tab = np.loadtxt(path, ndmin=2)
    if not np.array_equal(tab[:, 2], tab[:, 3]):
        raise ValueError("Only unidirectional QTFs are supported for now.")
    wa, wb = 2. * np.pi / tab[:, 0], 2. * np.pi / tab[:, 1]             # the file lists periods
    w_grid, ia = np.unique(wa, return_inverse=True)
    w_grid_b, ib = np.unique(wb, return_inverse=True)
    if w_grid.shape != w_grid_b.shape or not np.array_equal(w_grid, w_grid_b):
        raise ValueError("Both frequency columns in the input QTF must contain the same values.")
    head_deg, ih = np.unique(tab[:, 2], return_inverse=True)
    heads = np.deg2rad(head_deg)
    dof = np.rint(tab[:, 4]).astype(int) - 1
    scale = rho * g * ULEN * np.where(dof >= 3, ULEN, 1)
    val = scale * (tab[:, 7] + 1j * tab[:, 8])
    n = len(w_grid)
    qtf = np.zeros([n, n, len(heads), nDOF], dtype=complex)
    off = ia != ib
    # rows that carry the same (i1, i2) as an earlier row's mirror must still win in file order: interleave the two
    # assignments of every row the way a row loop does (entry, then its mirror)
    rows = np.arange(len(tab))
    order = np.argsort(np.concatenate([2 * rows, 2 * rows[off] + 1]), kind="stable")
    i1 = np.concatenate([ia, ib[off]])[order]
    i2 = np.concatenate([ib, ia[off]])[order]
    hh = np.concatenate([ih, ih[off]])[order]
    dd = np.concatenate([dof, dof[off]])[order]
    vv = np.concatenate([val, np.conj(val[off])])[order]
    flat = np.ravel_multi_index((i1, i2, hh, dd), qtf.shape)
    # last occurrence wins: keep, for every target, the latest position
    last = np.full(qtf.size, -1, dtype=np.int64)
    np.maximum.at(last, flat, np.arange(len(flat)))
    keep = last[last >= 0]
    qtf.reshape(-1)[flat[keep]] = vv[keep]
    return heads, w_grid, qtf


def write_rao4(path, w, beta, Xi):
    """The WAMIT .4 motion-RAO dump of raft_fowt.py:2027-2040."""
    with open(path, "w") as f:
        for iDoF in range(Xi.shape[0]):
            for w1, x in zip(w, Xi[iDoF, :]):
                f.write(f"{2*np.pi/w1: 8.4e} {beta: 8.4e} {iDoF+1} {np.abs(x): 8.4e} {np.angle(x): 8.4e} {x.real: 8.4e} {x.imag: 8.4e}\n")
