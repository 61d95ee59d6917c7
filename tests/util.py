import copy

import numpy as np

from tests import standin


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


def rel_err(a, b):
    a = np.asarray(a)
    b = np.asarray(b)
    den = np.max(np.abs(b))
    return np.max(np.abs(a - b)) / (den if den > 0 else 1.0)


def case_from_fixture(c):
    case = {}
    for k, v in c["case"].items():
        case[k] = list(v) if isinstance(v, (list, np.ndarray)) else v
    return copy.deepcopy(case)


def load_model_fixture(name):
    fx = standin.load_fixture(name)
    return fx, standin.build_model(fx["model"])
