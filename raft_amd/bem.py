"""Potential-flow (BEM) coefficient ingestion: WAMIT-format ``.1`` / ``.3`` files -> A_BEM, B_BEM, X_BEM.

Upstream, FOWT.readHydro (raft/raft_fowt.py:1444-1509) delegates the file parsing to pyHAMS
(``pyhams.pyhams.read_wamit1`` / ``read_wamit3``; pyHAMS is an un-vendored, un-pinned dependency -- pyproject.toml /
environment.yml -- and is NOT installed here), then interpolates to the model frequencies, dimensionalises and moves
the excitation into the wave-heading frame.  This module provides

  * ``read_wamit1`` / ``read_wamit3``: parsers written from the published WAMIT numeric-output layout
    (WAMIT user manual, "Numeric output files": ``.1``: PER I J A(I,J) [B(I,J)];  ``.3``: PER BETA I MOD PHA RE IM)
    and from the contract FOWT.readHydro states for its inputs (raft_fowt.py:1455-1458: with TFlag the first column
    is a PERIOD; the set PER = -1 is the zero-frequency limit, the set PER = 0 the infinite-frequency limit, and they
    come first).  PARITY UNPINNED for the parsers themselves: there is no pyHAMS here to compare with; they are
    checked by a write -> read round trip and through the reference's own readHydro running on top of them
    (oracle/make_golden.py registers this module as the ``pyhams.pyhams`` stub).
  * ``write_wamit1`` / ``write_wamit3``: the inverse (used to create the committed test decks).
  * ``read_hydro(fowt)``: the host mirror of FOWT.readHydro (same attributes: BEM_headings, A_BEM, B_BEM, X_BEM).

The heading interpolation that turns X_BEM into F_BEM for every (design, case, heading) runs on the device
(``raftx_bem_excitation``, include/raftx.h).
"""
import numpy as np

from .rigid import translate_matrix_6to6


def _period_to_w(per):
    per = np.asarray(per, dtype=float)
    with np.errstate(divide="ignore"):
        w = np.where(per > 0, 2 * np.pi / np.where(per > 0, per, 1.0), np.where(per < 0, 0.0, np.inf))
    return w


def read_wamit1(path, TFlag=False):
    """(addedMass [6,6,nf], damping [6,6,nf], w [nf]) in file order of the first-column values.  Rows of the
    limiting sets (PER = -1: zero frequency, PER = 0: infinite frequency) carry no damping column."""
    keys, rows = [], {}
    with open(path) as f:
        for line in f:
            p = line.split()
            if len(p) < 4:
                continue
            per = float(p[0])
            if per not in rows:
                rows[per] = []
                keys.append(per)
            rows[per].append((int(p[1]), int(p[2]), float(p[3]), float(p[4]) if len(p) > 4 else 0.0))
    nf = len(keys)
    A = np.zeros((6, 6, nf))
    B = np.zeros((6, 6, nf))
    for i, per in enumerate(keys):
        for (r, c, a, b) in rows[per]:
            A[r - 1, c - 1, i] = a
            B[r - 1, c - 1, i] = b
    first = np.array(keys)
    return A, B, (_period_to_w(first) if TFlag else first)


def read_wamit3(path, TFlag=False):
    """(mod, phase, real, imag [nHead,6,nf], w [nf], headings [nHead]) in file order."""
    pers, heads, data = [], [], {}
    with open(path) as f:
        for line in f:
            p = line.split()
            if len(p) < 7:
                continue
            per, beta, i = float(p[0]), float(p[1]), int(p[2])
            if per not in pers:
                pers.append(per)
            if beta not in heads:
                heads.append(beta)
            data[(per, beta, i)] = tuple(float(x) for x in p[3:7])
    nf, nh = len(pers), len(heads)
    out = np.zeros((4, nh, 6, nf))
    for (per, beta, i), v in data.items():
        out[:, heads.index(beta), i - 1, pers.index(per)] = v
    first = np.array(pers)
    return out[0], out[1], out[2], out[3], (_period_to_w(first) if TFlag else first), list(heads)


def write_wamit1(path, w, A, B, A0=None, Ainf=None):
    """Period-based ``.1`` file: optional zero-frequency (PER = -1) and infinite-frequency (PER = 0) sets, then one set
    per frequency.  A, B [6,6,nf] non-dimensional as WAMIT writes them."""
    with open(path, "w") as f:
        for per, M in ((-1.0, A0), (0.0, Ainf)):
            if M is not None:
                for r in range(6):
                    for c in range(6):
                        f.write(" %13.6E %5d %5d %13.6E\n" % (per, r + 1, c + 1, M[r, c]))
        for i, wi in enumerate(w):
            for r in range(6):
                for c in range(6):
                    f.write(" %13.6E %5d %5d %13.6E %13.6E\n" % (2 * np.pi / wi, r + 1, c + 1, A[r, c, i], B[r, c, i]))


def write_wamit3(path, w, headings, X):
    """Period-based ``.3`` file from complex X [nHead,6,nf] (non-dimensional)."""
    with open(path, "w") as f:
        for i, wi in enumerate(w):
            for ih, h in enumerate(headings):
                for j in range(6):
                    x = X[ih, j, i]
                    f.write(" %13.6E %13.6E %5d %13.6E %13.6E %13.6E %13.6E\n"
                            % (2 * np.pi / wi, h, j + 1, abs(x), np.degrees(np.angle(x)), x.real, x.imag))


def added_mass_damping(path1, w, rho_water, r0):
    """A_BEM, B_BEM [6,6,nw] from a WAMIT ``.1`` file as FOWT.readHydro builds them (raft_fowt.py:1455,1469-1478): the
    sets after the two limiting ones interpolated to the model frequencies ``w`` (the zero-frequency added mass / zero
    damping appended at w = 0), dimensionalised (rho A, rho w B) and moved by -r0 (translateMatrix6to6DOF)."""
    from scipy.interpolate import interp1d
    A, B, w1 = read_wamit1(path1, TFlag=True)
    w = np.asarray(w, dtype=float)
    Ai = interp1d(np.hstack([w1[2:], 0.0]), np.dstack([A[:, :, 2:], A[:, :, 0]]), assume_sorted=False, axis=2)(w)
    Bi = interp1d(np.hstack([w1[2:], 0.0]), np.dstack([B[:, :, 2:], np.zeros([6, 6])]), assume_sorted=False, axis=2)(w)
    shift = -np.asarray(r0, dtype=float)[:3]
    A_BEM = np.zeros([6, 6, len(w)])
    B_BEM = np.zeros([6, 6, len(w)])
    for iw in range(len(w)):
        A_BEM[:, :, iw] = translate_matrix_6to6(rho_water * Ai[:, :, iw], shift)
        B_BEM[:, :, iw] = translate_matrix_6to6(w[iw] * rho_water * Bi[:, :, iw], shift)
    return A_BEM, B_BEM


def read_hydro(fowt, path=None):
    """Host mirror of FOWT.readHydro (raft_fowt.py:1444-1509): sets BEM_headings, A_BEM, B_BEM [6,6,nw] and X_BEM
    [nHeadBEM,nDOF,nw] on ``fowt`` from ``path`` (default fowt.hydroPath) + '.1' / '.3'."""
    from scipy.interpolate import interp1d
    path = fowt.hydroPath if path is None else path
    M, P, R, Im, w3, heads = read_wamit3(path + ".3", TFlag=True)
    heads = np.array(heads) % 360
    order = np.argsort(heads)
    fowt.BEM_headings = heads[order]
    R, Im = R[order], Im[order]
    w = np.asarray(fowt.w)
    nw = len(w)
    Ri = interp1d(np.hstack([w3, 0.0]), np.dstack([R, np.zeros([len(heads), 6])]), assume_sorted=False, axis=2)(w)
    Ii = interp1d(np.hstack([w3, 0.0]), np.dstack([Im, np.zeros([len(heads), 6])]), assume_sorted=False, axis=2)(w)
    node = fowt.nodeList[fowt.reducedDOF[0][0]]
    fowt.A_BEM = np.zeros([fowt.nDOF, fowt.nDOF, nw])
    fowt.B_BEM = np.zeros([fowt.nDOF, fowt.nDOF, nw])
    fowt.A_BEM[:6, :6, :], fowt.B_BEM[:6, :6, :] = added_mass_damping(path + ".1", w, fowt.rho_water, node.r0)
    # dimensional excitation per unit amplitude, rotated into the frame of its own wave heading (surge / roll along the
    # waves: magnitudes then interpolate smoothly between headings, raft_fowt.py:1480-1496), moved to the node position
    Xg = fowt.rho_water * fowt.g * (Ri + 1j * Ii)                      # [nHeadBEM, 6, nw], global frame
    ang = np.radians(fowt.BEM_headings)[:, None, None]
    c, s = np.cos(ang), np.sin(ang)
    Xh = Xg.copy()
    Xh[:, [0, 3], :] = c * Xg[:, [0, 3], :] + s * Xg[:, [1, 4], :]        # (surge, roll) along the waves
    Xh[:, [1, 4], :] = -s * Xg[:, [0, 3], :] + c * Xg[:, [1, 4], :]       # (sway, pitch) across
    off = -np.asarray(node.r[:3], dtype=float)
    Xh[:, 3:, :] += np.cross(off[None, None, :], np.moveaxis(Xh[:, :3, :], 1, 2)).transpose(0, 2, 1)   # transformForce(offset=-node.r)
    fowt.X_BEM = np.zeros((Xh.shape[0], fowt.nDOF, nw), dtype=complex)
    fowt.X_BEM[:, :6, :] = Xh
    for name in ("A_BEM", "B_BEM", "X_BEM"):
        if np.isnan(getattr(fowt, name)).any():
            raise Exception("NaN values detected in HAMS calculations for %s. Check the geometry."
                            % {"A_BEM": "added mass", "B_BEM": "damping", "X_BEM": "excitation"}[name])
    return fowt
