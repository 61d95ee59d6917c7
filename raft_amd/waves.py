"""Host-side sea-state set-up: the feeders of the hot path that stay in Python.

Restates (it does not import) the reference's
    raft/helpers.py:703-760    JONSWAP
    raft/helpers.py:377-392    waveNumber  (fixed-point, rel. tol 1e-3 -- the
                               kernels always consume THIS k, never re-solve)
    raft/helpers.py:828-906    getFromDict (the 1-D/tile subset used for cases)
    raft/raft_fowt.py:1742-1774  sea-state block of FOWT.calcHydroExcitation
    raft/helpers.py:762-784, 687-700  getRAO, getPSD
"""
import numpy as np


def wave_number(omega, h, e=0.001):
    """helpers.py:377-392 -- the reference's loose fixed-point dispersion solve."""
    g = 9.81
    k1 = omega * omega / g
    k2 = omega * omega / (np.tanh(k1 * h) * g)
    while np.abs(k2 - k1) / k1 > e:
        k1 = k2
        k2 = omega * omega / (np.tanh(k1 * h) * g)
    return k2


def frequency_grid(min_freq, max_freq):
    """raft_model.py:56-57: w = arange(min, max+0.5*min, min)*2*pi."""
    return np.arange(min_freq, max_freq + 0.5 * min_freq, min_freq) * 2 * np.pi


def jonswap(ws, Hs, Tp, Gamma=None):
    """helpers.py:703-760."""
    if not Gamma:
        TpOvrSqrtHs = Tp / np.sqrt(Hs)
        if TpOvrSqrtHs <= 3.6:
            Gamma = 5.0
        elif TpOvrSqrtHs >= 5.0:
            Gamma = 1.0
        else:
            Gamma = np.exp(5.75 - 1.15 * TpOvrSqrtHs)
    ws = np.array(ws) if isinstance(ws, (list, tuple, np.ndarray)) else np.array([ws])
    f = 0.5 / np.pi * ws
    fpOvrf4 = pow((Tp * f), -4.0)
    Cn = 1.0 - (0.287 * np.log(Gamma))
    Sigma = 0.07 * (f <= 1.0 / Tp) + 0.09 * (f > 1.0 / Tp)
    Alpha = np.exp(-0.5 * ((f * Tp - 1.0) / Sigma) ** 2)
    return 0.5 / np.pi * Cn * 0.3125 * Hs * Hs * fpOvrf4 / f * np.exp(-1.25 * fpOvrf4) * Gamma ** Alpha


def case_entry(case, key, n, dtype=float, default=None):
    """getFromDict(case, key, shape=n, dtype, default) -- helpers.py:828-906."""
    if key in case:
        val = case[key]
        if np.isscalar(val):
            return np.tile(dtype(val), n)
        if len(val) == n:
            return np.array([dtype(v) for v in val])
        raise ValueError(f"Value for key '{key}' is not the expected size of {n} and is instead: {val}")
    if default is None:
        raise ValueError(f"Key '{key}' not found in input file...")
    return np.tile(default, n)


def sea_state(case, w, dw):
    """raft_fowt.py:1742-1774.  Mutates ``case`` exactly as the reference does
    (entries become arrays) and returns (nWaves, beta[rad], S, zeta)."""
    nw = len(w)
    if np.isscalar(case['wave_heading']):
        nWaves = 1
    else:
        nWaves = len(case['wave_heading'])
    case['wave_heading'] = case_entry(case, 'wave_heading', nWaves, float, default=0)
    case['wave_spectrum'] = case_entry(case, 'wave_spectrum', nWaves, str, default='JONSWAP')
    case['wave_period'] = case_entry(case, 'wave_period', nWaves, float)
    case['wave_height'] = case_entry(case, 'wave_height', nWaves, float)
    case['wave_gamma'] = case_entry(case, 'wave_gamma', nWaves, float, default=0)

    beta = np.deg2rad(case['wave_heading'])
    zeta = np.zeros([nWaves, nw])
    S = np.zeros([nWaves, nw])
    for ih in range(nWaves):
        spec = case['wave_spectrum'][ih]
        if spec == 'unit':
            S[ih, :] = np.tile(1, nw)
            zeta[ih, :] = np.sqrt(2 * S[ih, :] * dw)
        elif spec == 'constant':
            S[ih, :] = case['wave_height'][ih]
            zeta[ih, :] = np.sqrt(2 * S * dw)       # (sic) raft_fowt.py:1766 broadcasts all rows
        elif spec == 'JONSWAP':
            S[ih, :] = jonswap(w, case['wave_height'][ih], case['wave_period'][ih],
                               Gamma=case['wave_gamma'][ih])
            zeta[ih, :] = np.sqrt(2 * S[ih, :] * dw)
        elif spec in ['none', 'still']:
            zeta[ih, :] = np.zeros(nw)
            S[ih, :] = np.zeros(nw)
        else:
            raise ValueError(f"Wave spectrum input '{spec}' not recognized.")
    return nWaves, beta, S, zeta


def get_rao(Xi, zeta):
    """helpers.py:762-784."""
    if len(zeta.shape) != 1:
        raise Exception("zeta must be a 1D array")
    if Xi.shape[-1] != len(zeta):
        raise Exception("The last dimension of Xi must be the same length as zeta")
    idx = np.where(np.abs(zeta) > 1e-6)
    RAO = np.zeros_like(Xi, dtype=complex)
    RAO[..., idx] = Xi[..., idx] / zeta[idx]
    return RAO


def get_psd(xi, dw):
    """helpers.py:687-700."""
    if len(xi.shape) == 1:
        return 0.5 * np.abs(xi) ** 2 / dw
    if len(xi.shape) == 2:
        return np.sum(0.5 * np.abs(xi) ** 2 / dw, axis=0)
    raise Exception("getPSD must be passed an array with 1 or 2 dimensions.")
