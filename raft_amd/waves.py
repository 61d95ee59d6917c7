"""Host-side sea-state set-up: the scalar-per-run feeders of the hot path that stay in Python.

What the reference computes in
    raft/helpers.py:703-760      JONSWAP spectrum
    raft/helpers.py:377-392      waveNumber (a loose fixed-point dispersion solve, relative tolerance 1e-3)
    raft/raft_fowt.py:1742-1774  the sea-state block of FOWT.calcHydroExcitation (case entries -> S, zeta, beta)
    raft/helpers.py:762-784, 687-700  getRAO, getPSD
is restated here.  Two things are parity-critical and therefore keep the reference's ARITHMETIC ORDER (not its code):
the wave numbers -- the kernels always consume this loosely converged k, never a re-solved one -- and the wave
amplitudes zeta = sqrt(2 S dw), whose spectrum is a left-to-right product whose rounding the goldens carry.
"""
import numpy as np

GRAVITY_OF_DISPERSION = 9.81          # hard-wired upstream (helpers.py:379), independent of the site's g


def wave_number(omega, depth, rel_tol=0.001):
    """Dispersion relation by the reference's own iteration: start from the deep-water value, re-apply
    k <- w^2 / (g tanh(k h)) until two successive iterates agree to ``rel_tol`` (helpers.py:377-392).  The result is
    deliberately NOT the converged root: it is what every reference array downstream was computed with."""
    w2_over_g = omega * omega / GRAVITY_OF_DISPERSION
    prev = w2_over_g
    cur = omega * omega / (np.tanh(prev * depth) * GRAVITY_OF_DISPERSION)
    while np.abs(cur - prev) / prev > rel_tol:
        prev, cur = cur, omega * omega / (np.tanh(cur * depth) * GRAVITY_OF_DISPERSION)
    return cur


def frequency_grid(min_freq, max_freq):
    """Model frequencies [rad/s] from the YAML settings in Hz (raft_model.py:56-57)."""
    return np.arange(min_freq, max_freq + 0.5 * min_freq, min_freq) * 2 * np.pi


def peak_enhancement(Hs, Tp):
    """IEC 61400-3 default of the JONSWAP peak-shape parameter from the steepness measure Tp / sqrt(Hs)."""
    steep = Tp / np.sqrt(Hs)
    if steep <= 3.6:
        return 5.0
    if steep >= 5.0:
        return 1.0
    return np.exp(5.75 - 1.15 * steep)


def jonswap(omega, Hs, Tp, Gamma=None):
    """One-sided JONSWAP spectrum S(w) [m^2/(rad/s)] (helpers.py:703-760); ``Gamma`` falsy -> the IEC default.  The
    final product is evaluated left to right in the reference's factor order (bit-level parity of zeta)."""
    gamma = Gamma if Gamma else peak_enhancement(Hs, Tp)
    omega = np.atleast_1d(np.asarray(omega, dtype=float))
    hz = 0.5 / np.pi * omega
    rel4 = pow((Tp * hz), -4.0)                                  # (f_p / f)^4
    norm = 1.0 - (0.287 * np.log(gamma))
    width = 0.07 * (hz <= 1.0 / Tp) + 0.09 * (hz > 1.0 / Tp)
    peak = np.exp(-0.5 * ((hz * Tp - 1.0) / width) ** 2)
    return 0.5 / np.pi * norm * 0.3125 * Hs * Hs * rel4 / hz * np.exp(-1.25 * rel4) * gamma ** peak


def case_entry(case, key, n, dtype=float, default=None):
    """One entry of a load-case row as a length-n array: scalars are repeated, sequences must have n items, a missing
    key takes ``default`` (the 1-D subset of the reference's YAML accessor, helpers.py:828-906, with its messages)."""
    if key not in case:
        if default is None:
            raise ValueError(f"Key '{key}' not found in input file...")
        return np.tile(default, n)
    val = case[key]
    if np.isscalar(val):
        return np.tile(dtype(val), n)
    if len(val) != n:
        raise ValueError(f"Value for key '{key}' is not the expected size of {n} and is instead: {val}")
    return np.array([dtype(v) for v in val])


def _spectrum_row(kind, Hs, Tp, gamma, w):
    """S(w) of one wave train, or None for the kinds the caller special-cases."""
    if kind == 'unit':
        return np.tile(1, len(w)).astype(float)
    if kind == 'JONSWAP':
        return jonswap(w, Hs, Tp, Gamma=gamma)
    if kind in ('none', 'still'):
        return np.zeros(len(w))
    return None


def sea_state(case, w, dw):
    """(nWaves, beta [rad], S [nWaves,nw], zeta [nWaves,nw]) of a load case (raft_fowt.py:1742-1774).  As upstream, the
    case's wave entries are REPLACED by per-wave-train arrays (callers rely on that side effect)."""
    n_trains = 1 if np.isscalar(case['wave_heading']) else len(case['wave_heading'])
    for key, kind, dflt in (('wave_heading', float, 0), ('wave_spectrum', str, 'JONSWAP'), ('wave_period', float, None),
                            ('wave_height', float, None), ('wave_gamma', float, 0)):
        case[key] = case_entry(case, key, n_trains, kind, default=dflt)
    S = np.zeros([n_trains, len(w)])
    zeta = np.zeros([n_trains, len(w)])
    for i, kind in enumerate(case['wave_spectrum']):
        if kind == 'constant':
            S[i, :] = case['wave_height'][i]
            zeta[i, :] = np.sqrt(2 * S * dw)       # (sic) raft_fowt.py:1766 takes the root of ALL rows: needs n_trains == 1
            continue
        row = _spectrum_row(kind, case['wave_height'][i], case['wave_period'][i], case['wave_gamma'][i], w)
        if row is None:
            raise ValueError(f"Wave spectrum input '{kind}' not recognized.")
        S[i, :] = row
        zeta[i, :] = np.sqrt(2 * S[i, :] * dw)
    return n_trains, np.deg2rad(case['wave_heading']), S, zeta


def get_rao(Xi, zeta):
    """Response per unit wave amplitude along the last axis; bins with |zeta| <= 1e-6 give zero (helpers.py:762-784)."""
    zeta = np.asarray(zeta)
    if zeta.ndim != 1:
        raise Exception("zeta must be a 1D array")
    if Xi.shape[-1] != zeta.shape[0]:
        raise Exception("The last dimension of Xi must be the same length as zeta")
    live = np.abs(zeta) > 1e-6
    out = np.zeros_like(Xi, dtype=complex)
    out[..., live] = Xi[..., live] / zeta[live]
    return out


def get_psd(xi, dw):
    """One-sided power spectral density of complex amplitudes; a 2-D input is summed over its first axis
    (independent excitation sources), helpers.py:687-700."""
    if xi.ndim == 1:
        return 0.5 * np.abs(xi) ** 2 / dw
    if xi.ndim == 2:
        return np.sum(0.5 * np.abs(xi) ** 2 / dw, axis=0)
    raise Exception("getPSD must be passed an array with 1 or 2 dimensions.")


def wave_kin(zeta0, beta, w, k, h, r, rho=1025.0, g=9.81):
    """Wave velocity, acceleration and dynamic pressure amplitudes at one point r for one wave train -- helpers.getWaveKin
    (raft/helpers.py:187-236) over all frequencies at once, with its three depth regimes (k == 0: the "ill-conditioned"
    constants; k h > 89.4: deep-water exponentials; otherwise the hyperbolic ratios).  u, ud [3,nw], pDyn [nw]."""
    zeta0 = np.asarray(zeta0)
    w = np.asarray(w, dtype=float)
    k = np.asarray(k, dtype=float)
    nw = len(w)
    u = np.zeros((3, nw), dtype=complex)
    ud = np.zeros((3, nw), dtype=complex)
    pDyn = np.zeros(nw, dtype=complex)
    z = float(r[2])
    if z > 0:
        return u, ud, pDyn
    zeta = zeta0 * np.exp(-1j * (k * (np.cos(beta) * r[0] + np.sin(beta) * r[1])))
    zero, deep = k == 0.0, k * h > 89.4
    mid = ~(zero | deep)
    sh = np.ones(nw)
    ch = np.full(nw, 99999.0)
    cc = np.full(nw, 99999.0)
    sh[deep] = ch[deep] = np.exp(k[deep] * z)
    cc[deep] = np.exp(k[deep] * z) + np.exp(-k[deep] * (z + 2.0 * h))
    sh[mid] = np.sinh(k[mid] * (z + h)) / np.sinh(k[mid] * h)
    ch[mid] = np.cosh(k[mid] * (z + h)) / np.sinh(k[mid] * h)
    cc[mid] = np.cosh(k[mid] * (z + h)) / np.cosh(k[mid] * h)
    u[0] = w * zeta * ch * np.cos(beta)
    u[1] = w * zeta * ch * np.sin(beta)
    u[2] = 1j * w * zeta * sh
    ud[:] = 1j * w * u
    pDyn[:] = rho * g * zeta * cc
    return u, ud, pDyn
