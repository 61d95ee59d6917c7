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


def case_from_fixture(c):
    case = {}
    for k, v in c["case"].items():
        case[k] = list(v) if isinstance(v, (list, np.ndarray)) else v
    return copy.deepcopy(case)


def load_model_fixture(name):
    fx = standin.load_fixture(name)
    return fx, standin.build_model(fx["model"])


# ------------------------------------------------------------------ synthetic inputs
def random_strips(rng, S, nw=0, mcf_frac=0.0):
    """Seeded synthetic strip table: random orthonormal triads, mixed
    circular/rectangular strips, realistic magnitudes."""
    from raft_amd.strips import StripTable, NFIELD
    from raft_amd import strips as st
    rec = np.zeros((S, NFIELD))
    cms = []
    for s in range(S):
        A = rng.normal(size=(3, 3))
        Q, _ = np.linalg.qr(A)
        if np.linalg.det(Q) < 0:
            Q[:, 2] *= -1
        r = np.array([rng.uniform(-40, 40), rng.uniform(-40, 40), rng.uniform(-30, -0.3)])
        rec[s, st.F_X:st.F_X + 3] = r
        rec[s, st.F_AX:st.F_AX + 3] = r - np.array([1.0, -2.0, 0.5])
        rec[s, st.F_Q:st.F_Q + 3] = Q[:, 0]
        rec[s, st.F_P1:st.F_P1 + 3] = Q[:, 1]
        rec[s, st.F_P2:st.F_P2 + 3] = Q[:, 2]
        rho_v = rng.uniform(1e4, 4e5)
        rec[s, st.F_IQ] = rng.uniform(0, 1e5)
        rec[s, st.F_IP1] = rho_v * rng.uniform(1.5, 2.0)
        rec[s, st.F_IP2] = rho_v * rng.uniform(1.5, 2.0)
        rec[s, st.F_AI] = rng.uniform(-30, 30)
        rec[s, st.F_DQ:st.F_DEND + 1] = rng.uniform(0, 4e4, size=4)
        rec[s, st.F_CIRC] = float(rng.integers(0, 2))
        rec[s, st.F_MCF] = -1.0
        rec[s, st.F_RHOV] = rho_v
        if nw and rng.uniform() < mcf_frac:
            rec[s, st.F_MCF] = float(len(cms))
            rec[s, st.F_IP1] = rec[s, st.F_IP2] = 0.0
            cms.append((rng.uniform(1.2, 2.2, size=(2, nw)) + 1j * rng.uniform(-0.5, 0.5, size=(2, nw))))
    return StripTable(rec, np.array(cms) if cms else None)


def random_matrices(rng, nD, nw=0, freq_dep=False):
    M0 = np.zeros((nD, 6, 6))
    B0 = np.zeros((nD, 6, 6))
    C0 = np.zeros((nD, 6, 6))
    for d in range(nD):
        m = rng.uniform(1.5e7, 3e7)
        M0[d] = np.diag([m, m, m, m * 1500, m * 1500, m * 900])
        M0[d, 0, 4] = M0[d, 4, 0] = -m * 8.0
        M0[d, 1, 3] = M0[d, 3, 1] = m * 8.0
        B0[d] = np.diag(rng.uniform(1e4, 1e5, size=6)) * np.array([1, 1, 1, 1e3, 1e3, 1e3])
        B0[d, 3, 4], B0[d, 4, 3] = 2e6, -2e6            # gyroscopic-like antisymmetric part
        C0[d] = np.diag([7e4, 7e4, 4e6, 2e9, 2e9, 1e8]) * rng.uniform(0.8, 1.2)
        C0[d, 2, 4] = C0[d, 4, 2] = 1e5
    MBw = None
    if freq_dep:
        MBw = rng.uniform(0, 1, size=(nD, 2, 6, 6, nw)) * np.array([1e5, 1e4])[None, :, None, None, None]
    return M0, B0, C0, MBw


def synthetic_cases(rng, nC, nH, nw, depth=200.0, wmin=0.05, wmax=2.0):
    from raft_amd import waves
    w = np.linspace(wmin, wmax, nw) if nw > 1 else np.array([0.7])
    k = np.array([waves.wave_number(x, depth) for x in w])
    dw = (w[1] - w[0]) if nw > 1 else 0.1
    zeta = np.zeros((nC, nH, nw))
    beta = rng.uniform(0, 2 * np.pi, size=(nC, nH))
    for c in range(nC):
        for h in range(nH):
            S = waves.jonswap(w, rng.uniform(1, 10), rng.uniform(6, 16))
            zeta[c, h] = np.sqrt(2 * S * dw)
    return w, k, zeta, beta


# ------------------------------------------------------------------ C3 workload
def volturnus_sweep(base_design, scales, heading_adjust=0.0):
    """Member descriptors of the C3 sweep, vectorised over designs: the five parameters of
    raft/parametersweep.py:33-37 (centre-column d, outer-column d, draft, outer-column radius, pontoon height)
    times ``scales`` [nD,5], with the dependent-geometry edits of :56-87 -- the same edits
    oracle/make_golden.py:volturnus_variant applies to the design dict, here applied straight to the descriptor
    arrays (no per-design Python).  ``base_design``: examples/VolturnUS-S_example.yaml (members: centre column,
    outer column x3, pontoon x3, upper beam x3, tower)."""
    from raft_amd import geometry as G
    scales = np.asarray(scales, dtype=float)
    nD = len(scales)
    base = G.describe_unit(base_design, heading_adjust=heading_adjust)
    heads = [np.atleast_1d(np.array(m.get("heading", 0.0), dtype=float)) for m in base_design["platform"]["members"]]
    assert [len(h) for h in heads] == [1, 3, 3, 3], "not the VolturnUS-S member layout"
    sw = G.SweepTables(base, nD)
    ccD, ocD, T, ocR, pH = 10.0 * scales[:, 0], 12.5 * scales[:, 1], -20.0 * scales[:, 2], 51.75 * scales[:, 3], 7.0 * scales[:, 4]
    z0 = np.zeros(nD)
    col = lambda *xs: np.stack([np.broadcast_to(np.asarray(x, dtype=float), (nD,)) for x in xs], axis=1)
    sw.set_ends(0, col(z0, z0, T), col(z0, z0, 15.0), heading=heads[0][0] + heading_adjust)
    sw.set_diameter(0, ccD)
    for c in range(3):
        h = heads[1][c] + heading_adjust
        sw.set_ends(1 + c, col(ocR, z0, T), col(ocR, z0, 15.0), heading=h)
        sw.set_diameter(1 + c, ocD)
        sw.set_ends(4 + c, col(ccD / 2, z0, T + pH / 2), col(ocR - ocD / 2, z0, T + pH / 2), heading=heads[2][c] + heading_adjust)
        sw.set_diameter(4 + c, np.full(nD, 12.4), pH)
        sw.set_ends(7 + c, col(ccD / 2, z0, 14.545), col(ocR - ocD / 2, z0, 14.545), heading=heads[3][c] + heading_adjust)
    return sw


# ------------------------------------------------------------------ moorMod == 2 stand-in
class FakeLines:
    """Stand-in for the MoorPy system of a ``moorMod == 2`` unit (MoorPy itself is absent here): the same call
    surface the reference uses (raft_model.py:1023-1030,1069-1072; raft_fowt.py:2281-2289) with matrices that
    depend on the motion amplitudes handed to updateMooringDynamicMatrices, the way the lumped-mass line drag does."""

    class _Body:
        def __init__(self, r6):
            self.r6 = np.asarray(r6, dtype=float)

    def __init__(self, r6, w):
        self.bodyList = [self._Body(r6)]
        self.lineList = []
        self.w = np.asarray(w)
        self.level = 0.0
        self.calls = 0

    def update(self, Xi, S):
        self.calls += 1
        v = self.w[None, :] * np.abs(np.asarray(Xi)[:3])
        self.level = float(np.sqrt(np.sum(v ** 2 * (1.0 + S[None, :]))))

    def getCoupledDynamicMatrices(self, lines_only=True):
        assert lines_only
        sym = lambda a: 0.5 * (a + a.T)
        rng = np.random.default_rng(7)
        M = sym(rng.uniform(0, 1, (6, 6))) * 2e4 + np.diag([3e5, 3e5, 2e5, 1e7, 1e7, 2e7])
        A = sym(rng.uniform(0, 1, (6, 6))) * 1e4 + np.diag([1e5, 1e5, 1e5, 5e6, 5e6, 5e6])
        B = (sym(rng.uniform(0, 1, (6, 6))) * 2e4 + np.diag([4e5, 4e5, 2e5, 3e7, 3e7, 6e7])) * (0.2 + self.level)
        C = sym(rng.uniform(0, 1, (6, 6))) * 1e3 + np.diag([8e4, 8e4, 2e4, 2e8, 2e8, 1.5e8])
        return M, A, B, C


def attach_fake_lines(model):
    for f in model.fowtList:
        if not hasattr(f, "nodeList"):                      # stand-in units (tests/standin.py): PRP-referred rigid body
            node = standin.Obj()
            node.r = np.array([f.x_ref, f.y_ref, 0.0])
            f.nodeList, f.reducedDOF, f.r6 = [node], [[0, 0]], np.r_[node.r, 0.0, 0.0, 0.0]
        f.ms = FakeLines(np.r_[f.r6[:3] + np.array([0.3, -0.2, -1.5]), 0, 0, 0], f.w)
        f.moorMod = 2
        f.updateMooringDynamicMatrices = (lambda Xi, S, ms=f.ms: ms.update(Xi, S))
