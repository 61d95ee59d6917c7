import copy

import numpy as np

from raft_amd import snapshot as standin


from raft_amd.metrics import group_rel_err, rao_group_err, psd_group_err, rel_err     # noqa: F401,E402
from raft_amd.geometry import volturnus_sweep                                            # noqa: F401,E402


from raft_amd.snapshot import case_from_fixture, ref_headings, load_model_fixture     # noqa: F401,E402


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


class FakeStaticLines:
    """Stand-in for the MoorPy system of a quasi-static (``moorMod == 0``) unit in FOWT.saveTurbineOutputs
    (raft_fowt.py:2356-2399): the two calls the method makes -- the coupled stiffness with the tension Jacobian
    [2 nLines, 6] and the mean line-end tensions -- and a line list of the right length."""

    def __init__(self, n_lines=3, seed=3):
        rng = np.random.default_rng(seed)
        self.lineList = [object() for _ in range(n_lines)]
        self.J = rng.uniform(-1.0, 1.0, (2 * n_lines, 6)) * np.array([4e4, 4e4, 9e4, 2e5, 2e5, 5e4])
        self.T = rng.uniform(1.5e6, 2.5e6, 2 * n_lines)
        self.C = np.diag([7e4, 7e4, 0.0, 0.0, 0.0, 1e8])

    def getCoupledStiffness(self, lines_only=True, tensions=False):
        assert lines_only
        return (self.C.copy(), self.J.copy()) if tensions else self.C.copy()

    def getTensions(self):
        return self.T.copy()


def attach_fake_lines(model):
    for f in model.fowtList:
        if not hasattr(f, "nodeList"):                      # stand-in units (raft_amd/snapshot.py): PRP-referred rigid body
            node = standin.Obj()
            node.r = np.array([f.x_ref, f.y_ref, 0.0])
            f.nodeList, f.reducedDOF, f.r6 = [node], [[0, 0]], np.r_[node.r, 0.0, 0.0, 0.0]
        f.ms = FakeLines(np.r_[f.r6[:3] + np.array([0.3, -0.2, -1.5]), 0, 0, 0], f.w)
        f.moorMod = 2
        f.updateMooringDynamicMatrices = (lambda Xi, S, ms=f.ms: ms.update(Xi, S))


class FakeArrayLines:
    """Stand-in for the SHARED mooring system of an array with ``moorMod == 2`` (raft_model.py:1173-1182,1305-1316):
    dense symmetric 6N x 6N matrices whose damping depends on the motions of EVERY unit handed to
    Model.updateMooringDynamicMatrices (a list of nFOWT [6,nw] arrays + the heading-0 spectrum)."""

    def __init__(self, n_unit, w):
        self.n, self.w = 6 * n_unit, np.asarray(w)
        self.level, self.calls, self.seen = 0.0, 0, None

    def update(self, Xi_list, S):
        self.calls += 1
        self.seen = [np.array(x) for x in Xi_list]
        v = [self.w[None, :] * np.abs(np.asarray(x)[:3]) * (1.0 + 0.1 * i) for i, x in enumerate(Xi_list)]
        self.level = float(np.sqrt(sum(np.sum(a ** 2 * (1.0 + S[None, :])) for a in v)))

    def getCoupledDynamicMatrices(self, lines_only=True):
        assert lines_only
        sym = lambda a: 0.5 * (a + a.T)
        rng = np.random.default_rng(11)
        n = self.n
        blk = np.tile([3e5, 3e5, 2e5, 1e7, 1e7, 2e7], n // 6)
        M = sym(rng.uniform(0, 1, (n, n))) * 2e4 + np.diag(blk)
        A = sym(rng.uniform(0, 1, (n, n))) * 1e4 + np.diag(blk / 3)
        B = (sym(rng.uniform(0, 1, (n, n))) * 2e4 + np.diag(blk * 1.5)) * (0.2 + self.level)
        C = sym(rng.uniform(-1, 1, (n, n))) * 2e4 + np.diag(np.tile([8e4, 8e4, 2e4, 2e8, 2e8, 1.5e8], n // 6))
        return M, A, B, C


def attach_fake_array_lines(model):
    """Array-level ``moorMod == 2`` on a model (live reference Model or stand-in)."""
    model.ms = FakeArrayLines(len(model.fowtList), model.w)
    model.moorMod = 2
    model.updateMooringDynamicMatrices = (lambda Xi, S, ms=model.ms: ms.update(Xi, S))


def attach_fake_lines_at(model, arm):
    """FakeLines on the (single) unit of ``model`` with the mooring body ``arm`` away from the unit's reduced-DOF reference
    node (raft_model.py:1027) -- for stand-in units whose golden was made with that arm on the live object
    (tests/golden/flex_moormod2.npz: the reference's flexible deck)."""
    f = model.fowtList[0]
    node = standin.Obj()
    node.r = np.zeros(3)
    f.nodeList, f.reducedDOF = [node], [[0, 0]]
    f.ms = FakeLines(np.r_[np.asarray(arm, dtype=float), 0.0, 0.0, 0.0], f.w)
    f.moorMod = 2
    f.updateMooringDynamicMatrices = (lambda Xi, S, ms=f.ms: ms.update(Xi, S))
    return f.ms
