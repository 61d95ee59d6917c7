"""Parity metrics of SURVEY.md 8d (group-relative errors of responses, RAOs and motion PSDs)."""
import numpy as np


def group_rel_err(a, b):
    """SURVEY.md 8d parity metric: max|a-b| / max|b|, jointly over the
    translational DOFs and jointly over the rotational DOFs (never per-DOF:
    un-excited DOFs are round-off in the reference itself).  The DOF axis is
    the second-to-last axis; systems with 6N DOFs are grouped per unit."""
    a = np.asarray(a)
    b = np.asarray(b)
    n = a.shape[-2]
    errs = []
    for u in range(n // 6):
        for sl in (slice(6 * u, 6 * u + 3), slice(6 * u + 3, 6 * u + 6)):
            den = np.max(np.abs(b[..., sl, :]))
            num = np.max(np.abs(a[..., sl, :] - b[..., sl, :]))
            errs.append(num / den if den > 0 else num)
    return max(errs)


def rao_group_err(Xi_a, Xi_b, zeta):
    """SURVEY.md 8d parity metric proper: RAO = getRAO(Xi, zeta) (helpers.py:762-784: Xi / zeta where |zeta| > 1e-6,
    zero elsewhere), then max|RAO_a - RAO_b| / max|RAO_b| jointly over {surge, sway, heave} and over {roll, pitch, yaw}.
    Xi_* [..., 6, nw], zeta [nw]."""
    from raft_amd import waves
    return group_rel_err(waves.get_rao(np.asarray(Xi_a), np.asarray(zeta)), waves.get_rao(np.asarray(Xi_b), np.asarray(zeta)))


def psd_group_err(Xi_a, Xi_b, dw):
    """The same on the motion PSDs (getPSD, helpers.py:687-700), Xi_* [nHead, 6, nw]."""
    pa = np.sum(0.5 * np.abs(np.asarray(Xi_a)) ** 2 / dw, axis=0)
    pb = np.sum(0.5 * np.abs(np.asarray(Xi_b)) ** 2 / dw, axis=0)
    return group_rel_err(pa, pb)


def rel_err(a, b):
    a = np.asarray(a)
    b = np.asarray(b)
    den = np.max(np.abs(b))
    return np.max(np.abs(a - b)) / (den if den > 0 else 1.0)
