"""The reference's helper-level known answers (tests/test_helpers.py:41-69: waveNumber, getWaveKin) driven through the
C-ABI: a one-strip design whose inertia coefficients pick out the wave acceleration, and one whose end area picks out
the dynamic pressure.  Literals below are the ones printed in the reference's test file (rtol 1e-5 there and here)."""
import numpy as np
import pytest

from raft_amd import waves
from raft_amd.strips import StripTable, NFIELD
from raft_amd import strips as st

W = np.array([0.1, 0.25, 0.5, 0.75])
ZETA0 = np.array([0.2, 0.2, 0.2, 0.2])
BETA, DEPTH = 30.0, 200.0                    # (sic) the reference's test passes 30 as radians
R = np.array([30.0, 45.0, -20.0])
DESIRED_K = np.array([0.00233623, 0.0071452, 0.02548611, 0.05733945])
DESIRED_UD = np.array([[-0.0000644885 + 0.0006909710j, -0.0005359019 + 0.0018317440j, -0.0039364177 + 0.0024438000j, -0.0041686415 - 0.0036067400j],
                       [0.0004130725 - 0.0044259010j, 0.0034326291 - 0.0117329200j, 0.0252140594 - 0.0156533200j, 0.0267015296 + 0.0231023400j],
                       [-0.0017800228 - 0.0001661310j, -0.0101901044 - 0.0029812600j, -0.0158396548 - 0.0255142000j, 0.0233821912 - 0.0270249700j]])
DESIRED_PDYN = np.array([1963.730340920 + 183.276331860j, 1703.156386190 + 498.282218140j, 637.171137130 + 1026.342526750j,
                         -417.980049950 + 483.098446900j])


def test_wave_number_literals():
    k = np.array([waves.wave_number(x, DEPTH) for x in W])
    np.testing.assert_allclose(k, DESIRED_K, rtol=1e-5)


def _strip(Iq, Ip1, Ip2, a_i):
    rec = np.zeros((1, NFIELD))
    rec[0, st.F_X:st.F_X + 3] = R
    rec[0, st.F_AX:st.F_AX + 3] = R
    rec[0, st.F_Q:st.F_Q + 3] = [0, 0, 1]
    rec[0, st.F_P1:st.F_P1 + 3] = [1, 0, 0]
    rec[0, st.F_P2:st.F_P2 + 3] = [0, 1, 0]
    rec[0, st.F_IQ], rec[0, st.F_IP1], rec[0, st.F_IP2], rec[0, st.F_AI] = Iq, Ip1, Ip2, a_i
    rec[0, st.F_CIRC], rec[0, st.F_MCF] = 1.0, -1.0
    return StripTable(rec)


def check_wave_kin(ctx):
    k = np.array([waves.wave_number(x, DEPTH) for x in W])
    eye = np.eye(6)[None]
    # design 0: unit inertia in every direction -> f3 = ud;  design 1: unit end area along q = e_z -> f3 = pDyn e_z
    ctx.upload_designs([_strip(1.0, 1.0, 1.0, 0.0), _strip(0.0, 0.0, 0.0, 1.0)], np.repeat(eye, 2, 0), np.zeros((2, 6, 6)),
                       np.repeat(eye, 2, 0), len(W))
    ctx.upload_cases(W, k, DEPTH, 1025.0, 9.81, ZETA0[None, None], np.array([[BETA]]))
    F = ctx.excitation()
    np.testing.assert_allclose(F[0, 0, 0, :3], DESIRED_UD, rtol=1e-5)
    np.testing.assert_allclose(F[1, 0, 0, 2], DESIRED_PDYN, rtol=1e-5)
    # moments: r x f3 (helpers.py:468-483, pinned upstream by test_translateForce3to6DOF)
    np.testing.assert_allclose(F[0, 0, 0, 3:], np.cross(R[None], DESIRED_UD.T).T, rtol=2e-5)


def test_oracle_wave_kinematics_literals(oracle_ctx):
    check_wave_kin(oracle_ctx)


@pytest.mark.gpu
def test_hip_wave_kinematics_literals(hip_ctx):
    check_wave_kin(hip_ctx)


# ------------------------------------------------------------------ rigid-shift helpers (tests/test_helpers.py:88-155)
FIN = np.array([0.5 + 3j, 2.0 + 1.5j, 3.0 + 0.7j])
DESIRED_F6 = np.array([0.5 + 3.0j, 2.0 + 1.5j, 3.0 + 0.7j, 0.0 - 3.1j, -1.5 + 8.3j, 1.0 - 4.5j])
M3 = np.array([[0.73, 2.41, 3.88], [1.25, 9.12, 5.79], [5.37, 7.94, 8.63]])
R36 = np.array([10.0, 20.0, 30.0])
DESIRED_M3TO6 = np.array([[7.300e-01, 2.410e+00, 3.880e+00, 5.300e+00, -1.690e+01, 9.500e+00],
                          [1.250e+00, 9.120e+00, 5.790e+00, -1.578e+02, -2.040e+01, 6.620e+01],
                          [5.370e+00, 7.940e+00, 8.630e+00, -6.560e+01, 7.480e+01, -2.800e+01],
                          [5.300e+00, -1.578e+02, -6.560e+01, 3.422e+03, 2.108e+03, -2.546e+03],
                          [-1.690e+01, -2.040e+01, 7.480e+01, 8.150e+02, -1.255e+03, 5.650e+02],
                          [9.500e+00, 6.620e+01, -2.800e+01, -1.684e+03, 1.340e+02, 4.720e+02]])
M6 = np.array([[0.57, 0.64, 0.88, 0.12, 0.34, 0.56], [2.03, -13.02, 8.00, 0.78, 0.90, 0.12], [1.11, -0.15, 0.10, 0.34, 0.56, 0.78],
               [0.12, 0.78, 0.34, 0.90, 0.12, 0.34], [0.34, 0.90, 0.56, 0.12, 0.34, 0.56], [0.56, 0.12, 0.78, 0.34, 0.56, 0.78]])
DESIRED_M6TO6 = np.array([[5.70000e-01, 6.40000e-01, 8.80000e-01, -1.48000e+00, 8.64000e+00, -4.44000e+00],
                          [2.03000e+00, -1.30200e+01, 8.00000e+00, 5.51380e+02, -1.82000e+01, -1.70680e+02],
                          [1.11000e+00, -1.50000e-01, 1.00000e-01, 6.84000e+00, 3.28600e+01, -2.29200e+01],
                          [-1.48000e+00, 5.51380e+02, 6.84000e+00, -1.64203e+04, 1.20352e+03, 4.66774e+03],
                          [8.64000e+00, -1.82000e+01, 3.28600e+01, -1.28480e+02, -6.44600e+01, 9.87600e+01],
                          [-4.44000e+00, -1.70680e+02, -2.29200e+01, 5.55574e+03, -3.45240e+02, -1.62722e+03]])


def test_oracle_rigid_shift_helpers_on_the_references_literals(oracle_lib):
    """translateForce3to6DOF (test_helpers.py:88-94) and translateMatrix3to6DOF (:129-144) replayed on the C oracle's
    own helpers (oracle-only exports of oracle/raftx_oracle.c)."""
    import ctypes as C
    L = oracle_lib.lib
    vp = C.c_void_p
    L.raftx_oracle_translate_force.argtypes = [vp, vp, vp]
    L.raftx_oracle_translate_force.restype = None
    L.raftx_oracle_translate_matrix_3to6.argtypes = [vp, vp, vp]
    L.raftx_oracle_translate_matrix_3to6.restype = None
    fin = np.ascontiguousarray(FIN.astype(np.complex128))
    r = np.array([1.0, 2.0, 3.0])
    out = np.zeros(6, dtype=np.complex128)
    L.raftx_oracle_translate_force(fin.ctypes.data, r.ctypes.data, out.ctypes.data)
    np.testing.assert_allclose(out, DESIRED_F6, rtol=1e-5, atol=0)
    m = np.ascontiguousarray(M3)
    out36 = np.zeros((6, 6))
    L.raftx_oracle_translate_matrix_3to6(m.ctypes.data, R36.ctypes.data, out36.ctypes.data)
    np.testing.assert_allclose(out36, DESIRED_M3TO6, rtol=1e-5, atol=0)


def test_host_translate_matrix_6to6_literals():
    """translateMatrix6to6DOF (test_helpers.py:147-155, as upstream: only the part it prints) on the host helper the
    drop-in and the WAMIT ingestion use (raft_amd/rigid.py)."""
    from raft_amd.rigid import translate_matrix_6to6
    np.testing.assert_allclose(translate_matrix_6to6(M6, R36), DESIRED_M6TO6, rtol=1e-5, atol=0)


# ------------------------------------------------------------------ getKinematics (tests/test_helpers.py:26-38)
KIN_R = np.array([2.0, 2.0, 2.0])
KIN_W = np.array([0.5, 0.75])
KIN_XI = np.array([[1, 2 + 1j], [0.1 + 0.2j, 0.3 + 0.4j], [0.5 + 0.6j, 0.7 + 0.8j], [0.9 + 1.0j, 1.1 + 1.2j], [1.3 + 1.4j, 1.5 + 1.6j],
                   [1.7 + 1.8j, 1.9 + 2.0j]])
KIN_V = np.array([[4.00000000e-01 + 0.1j, -1.50000000e-01 + 0.9j], [-9.00000000e-01 + 0.85j, -1.50000000e+00 + 1.425j],
                  [1.00000000e-01 - 0.15j, 1.66533454e-16 - 0.075j]])


def check_node_velocity(ctx):
    """The node velocity of getKinematics, v = i w (xi_t + theta x r), through the C-ABI: in still water the relative
    velocity of the drag linearisation IS minus that velocity, and for a strip with axial drag only along e_c
    B_drag[c, c] = (Bq + Bend) sqrt(1/2 sum_w |v_c|^2) (raft_member.py:2084-2110, helpers.py:684)."""
    k = np.array([waves.wave_number(x, DEPTH) for x in KIN_W])
    tabs = []
    triads = [((1, 0, 0), (0, 1, 0), (0, 0, 1)), ((0, 1, 0), (0, 0, 1), (1, 0, 0)), ((0, 0, 1), (1, 0, 0), (0, 1, 0))]
    for q, p1, p2 in triads:
        rec = np.zeros((1, NFIELD))
        rec[0, st.F_X:st.F_X + 3] = [0.0, 0.0, -5.0]
        rec[0, st.F_AX:st.F_AX + 3] = KIN_R
        rec[0, st.F_Q:st.F_Q + 3], rec[0, st.F_P1:st.F_P1 + 3], rec[0, st.F_P2:st.F_P2 + 3] = q, p1, p2
        rec[0, st.F_DQ] = 1.0
        rec[0, st.F_CIRC], rec[0, st.F_MCF] = 0.0, -1.0
        tabs.append(StripTable(rec))
    eye = np.repeat(np.eye(6)[None], 3, 0)
    ctx.upload_designs(tabs, eye, np.zeros((3, 6, 6)), eye, len(KIN_W))
    ctx.upload_cases(KIN_W, k, DEPTH, 1025.0, 9.81, np.zeros((1, 1, 2)), np.array([[0.0]]))
    B, _ = ctx.linearize(np.repeat(KIN_XI[None, None], 3, 0), want_F=False)
    for c in range(3):
        want = np.sqrt(0.5 * np.sum(np.abs(KIN_V[c]) ** 2))
        np.testing.assert_allclose(B[c, 0, c, c], want, rtol=1e-5)


def test_oracle_node_velocity_literals(oracle_ctx):
    check_node_velocity(oracle_ctx)


@pytest.mark.gpu
def test_hip_node_velocity_literals(hip_ctx):
    check_node_velocity(hip_ctx)
