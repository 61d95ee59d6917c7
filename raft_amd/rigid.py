"""Rigid-body shift algebra used on the host side of the drop-in (the device kernels fuse the same algebra into the
strip sweeps).  Pinned on the reference's own literals (tests/test_reference_helpers.py)."""
import numpy as np


def alternator(r):
    """H(r) with H @ v = v x r (raft/helpers.py:428-437 getH: the transpose of the usual cross-product matrix)."""
    return np.array([[0.0, r[2], -r[1]], [-r[2], 0.0, r[0]], [r[1], -r[0], 0.0]])


def translate_matrix_6to6(Min, r):
    """A 6 x 6 matrix referred to a point moved by r (raft/helpers.py:563-585 translateMatrix6to6DOF)."""
    Min = np.asarray(Min, dtype=float)
    H = alternator(r)
    tt, tr, rt, rr = Min[:3, :3], Min[:3, 3:], Min[3:, :3], Min[3:, 3:]
    out = np.empty((6, 6))
    out[:3, :3] = tt
    out[:3, 3:] = tt @ H + tr
    out[3:, :3] = out[:3, 3:].T
    out[3:, 3:] = H @ tt @ H.T + rt @ H + H.T @ tr + rr
    return out
