"""ctypes binding of the raftx C-ABI (include/raftx.h).

``RaftxLib(path)`` binds ANY shared object that implements the header; the
product only ever binds ``raft_amd/csrc/libraftx_hip.so`` (see backend.py).
The test-suite binds ``oracle/libraftx_oracle.so`` through the same class so
parity tests drive both implementations with identical calls.
"""
import ctypes as C
import os

import numpy as np

NFIELD = 32
FLAG_CONVERGED = 1
FLAG_NAN = 2

_c_double_p = C.POINTER(C.c_double)
_c_i64_p = C.POINTER(C.c_int64)
_c_i32_p = C.POINTER(C.c_int32)
_vp = C.c_void_p

EXPORTS = (
    "raftx_version", "raftx_is_device", "raftx_ctx_create", "raftx_ctx_destroy",
    "raftx_last_error", "raftx_upload_designs", "raftx_upload_cases",
    "raftx_excitation", "raftx_linearize", "raftx_solve_dynamics",
    "raftx_solve_system", "raftx_last_kernel_ms",
    "raftx_solve_dynamics_device", "raftx_fetch_results", "raftx_debug_math", "raftx_debug_math_table", "raftx_last_solve_kernel", "raftx_device_synchronize", "raftx_motion_stats", "raftx_solve_system_resident", "raftx_qtf_slender", "raftx_channel_stats", "raftx_qtf_force", "raftx_set_linearisation_point", "raftx_fetch_linearisation_point",
    "raftx_build_designs", "raftx_fetch_strips", "raftx_fetch_statics", "raftx_channel_stats_poly", "raftx_qtf_slender_rows", "raftx_bem_excitation", "raftx_qtf_kay", "raftx_host_alloc", "raftx_host_free", "raftx_device_locality", "raftx_solve_dense",
    "raftx_sweep_stats",
    "raftx_sweep_submit",
    "raftx_sweep_prepare",
    "raftx_sweep_launch",
    "raftx_sweep_wait", "raftx_sweep_solve_span", "raftx_sweep_generation",
    "raftx_sweep_cancel", "raftx_device_count", "raftx_solve_dense_batch", "raftx_dense_resident", "raftx_solve_dense_resident", "raftx_flex_solve", "raftx_flex_start", "raftx_debug_flex_gemm",
    "raftx_comm_unique_id", "raftx_comm_init", "raftx_comm_destroy", "raftx_comm_broadcast", "raftx_comm_gather_rows",
    "raftx_comm_gather_xi", "raftx_comm_reduce_sum",
    "raftx_variant_program", "raftx_expand_variants", "raftx_sweep_prepare_variants",
    "raftx_strip_kinematics", "raftx_strip_drag", "raftx_response_stats",
)
WANT_BDRAG, WANT_FWAVE, WANT_Z = 1, 2, 4


class RaftxError(RuntimeError):
    pass


def _f64(a, shape=None, name="array"):
    a = np.ascontiguousarray(a, dtype=np.float64)
    if shape is not None and tuple(a.shape) != tuple(shape):
        raise ValueError("%s has shape %s, expected %s" % (name, a.shape, tuple(shape)))
    return a


def _c128(a, shape=None, name="array"):
    a = np.ascontiguousarray(a, dtype=np.complex128)
    if shape is not None and tuple(a.shape) != tuple(shape):
        raise ValueError("%s has shape %s, expected %s" % (name, a.shape, tuple(shape)))
    return a


def _ptr(a):
    return None if a is None else a.ctypes.data_as(_vp)


class RaftxLib:
    def __init__(self, path):
        if not os.path.exists(path):
            raise RaftxError("shared library not found: %s" % path)
        self.path = path
        self.lib = C.CDLL(path)
        missing = [s for s in EXPORTS if not hasattr(self.lib, s)]
        if missing:
            raise RaftxError("%s does not export %s" % (path, missing))
        L = self.lib
        L.raftx_version.restype = C.c_int
        L.raftx_is_device.restype = C.c_int
        L.raftx_ctx_create.argtypes = [C.c_int, C.POINTER(_vp)]
        L.raftx_ctx_create.restype = C.c_int
        L.raftx_ctx_destroy.argtypes = [_vp]
        L.raftx_ctx_destroy.restype = None
        L.raftx_last_error.argtypes = [_vp]
        L.raftx_last_error.restype = C.c_char_p
        L.raftx_upload_designs.argtypes = [_vp, C.c_int, _vp, _vp, C.c_int, _vp, _vp, _vp,
                                           C.c_int, _vp, _vp, _vp]
        L.raftx_upload_designs.restype = C.c_int
        L.raftx_upload_cases.argtypes = [_vp, C.c_int, C.c_int, C.c_int, _vp, _vp,
                                         C.c_double, C.c_double, C.c_double, _vp, _vp]
        L.raftx_upload_cases.restype = C.c_int
        L.raftx_excitation.argtypes = [_vp, _vp]
        L.raftx_excitation.restype = C.c_int
        L.raftx_linearize.argtypes = [_vp, _vp, _vp, _vp]
        L.raftx_linearize.restype = C.c_int
        L.raftx_solve_dynamics.argtypes = [_vp, C.c_int, C.c_double, C.c_double, _vp,
                                           _vp, _vp, _vp, _vp, _vp, _vp]
        L.raftx_solve_dynamics.restype = C.c_int
        L.raftx_solve_system.argtypes = [_vp, C.c_int, C.c_int, C.c_int, C.c_int, _vp, _vp,
                                         _vp, _vp, _vp, _vp, _vp]
        L.raftx_solve_system.restype = C.c_int
        L.raftx_solve_dynamics_device.argtypes = [_vp, C.c_int, C.c_double, C.c_double, _vp, C.c_int]
        L.raftx_solve_dynamics_device.restype = C.c_int
        L.raftx_fetch_results.argtypes = [_vp, _vp, _vp, _vp, _vp, _vp, _vp]
        L.raftx_fetch_results.restype = C.c_int
        L.raftx_solve_system_resident.argtypes = [_vp, C.c_int, _vp, _vp, _vp, _vp]
        L.raftx_solve_system_resident.restype = C.c_int
        L.raftx_qtf_slender.argtypes = [_vp, C.c_int, C.c_int, _vp, _vp, C.c_double, C.c_double, C.c_double,
                                        _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]
        L.raftx_qtf_slender.restype = C.c_int
        L.raftx_qtf_slender_rows.argtypes = [_vp, C.c_int, C.c_int, _vp, _vp, C.c_double, C.c_double, C.c_double,
                                             _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, C.c_int, C.c_int, _vp]
        L.raftx_qtf_slender_rows.restype = C.c_int
        L.raftx_set_linearisation_point.argtypes = [_vp, _vp, C.c_int]
        L.raftx_set_linearisation_point.restype = C.c_int
        L.raftx_fetch_linearisation_point.argtypes = [_vp, _vp]
        L.raftx_fetch_linearisation_point.restype = C.c_int
        L.raftx_qtf_force.argtypes = [_vp, C.c_int, C.c_int, _vp, _vp, C.c_int, _vp, C.c_double, _vp, _vp, _vp]
        L.raftx_qtf_force.restype = C.c_int
        L.raftx_channel_stats.argtypes = [_vp, C.c_int, _vp, _vp, C.c_double, _vp, _vp]
        L.raftx_channel_stats.restype = C.c_int
        L.raftx_solve_dense.argtypes = [_vp, C.c_int, C.c_int, C.c_int, _vp, _vp, _vp, _vp, C.c_int, _vp, _vp, _vp]
        L.raftx_solve_dense.restype = C.c_int
        L.raftx_solve_dense_batch.argtypes = [_vp, C.c_int, C.c_int, C.c_int, C.c_int, _vp, _vp, _vp, _vp, C.c_int, _vp, _vp, _vp]
        L.raftx_solve_dense_batch.restype = C.c_int
        L.raftx_dense_resident.argtypes = [_vp, C.c_int, C.c_int, C.c_int, _vp, _vp, _vp, _vp, C.c_int]
        L.raftx_dense_resident.restype = C.c_int
        L.raftx_solve_dense_resident.argtypes = [_vp, C.c_int, _vp, C.c_int, _vp, _vp, _vp]
        L.raftx_solve_dense_resident.restype = C.c_int
        L.raftx_flex_solve.argtypes = [_vp, C.c_int, _vp, C.c_int, _vp, _vp, _vp, _vp, C.c_int, _vp, C.c_int, C.c_double, C.c_double,
                                       _vp, _vp, _vp, _vp, _vp, _vp]
        L.raftx_flex_solve.restype = C.c_int
        if hasattr(L, "raftx_flex_start"):
            L.raftx_flex_start.argtypes = [_vp, C.c_int, C.c_int, _vp]
            L.raftx_flex_start.restype = C.c_int
        L.raftx_debug_flex_gemm.argtypes = [_vp, C.c_int, C.c_int, _vp, _vp, _vp]
        L.raftx_debug_flex_gemm.restype = C.c_int
        L.raftx_device_locality.argtypes = [C.c_int, C.c_char_p, C.c_int, C.POINTER(C.c_int)]
        L.raftx_device_locality.restype = C.c_int
        L.raftx_host_alloc.argtypes = [_vp, C.c_size_t, C.POINTER(_vp)]
        L.raftx_host_alloc.restype = C.c_int
        L.raftx_host_free.argtypes = [_vp, _vp]
        L.raftx_host_free.restype = C.c_int
        L.raftx_qtf_kay.argtypes = [_vp, C.c_int, C.c_int, _vp, _vp, C.c_double, C.c_double, C.c_double, _vp, _vp, _vp, C.c_int, _vp]
        L.raftx_qtf_kay.restype = C.c_int
        L.raftx_bem_excitation.argtypes = [_vp, C.c_int, _vp, _vp, _vp, _vp, _vp, _vp]
        L.raftx_bem_excitation.restype = C.c_int
        L.raftx_channel_stats_poly.argtypes = [_vp, C.c_int, _vp, _vp, C.c_double, _vp, _vp]
        L.raftx_channel_stats_poly.restype = C.c_int
        L.raftx_motion_stats.argtypes = [_vp, C.c_double, _vp, _vp]
        L.raftx_motion_stats.restype = C.c_int
        L.raftx_debug_math.argtypes = [_vp, C.c_int, _vp, _vp, _vp, _vp]
        L.raftx_debug_math.restype = C.c_int
        L.raftx_sweep_launch.argtypes = [_vp, C.c_int]
        L.raftx_sweep_launch.restype = C.c_int
        L.raftx_device_synchronize.argtypes = [_vp]
        L.raftx_device_synchronize.restype = C.c_int
        L.raftx_last_solve_kernel.argtypes = [_vp, _vp, _vp, _vp]
        L.raftx_last_solve_kernel.restype = C.c_int
        L.raftx_debug_math_table.argtypes = [_vp, C.c_int, _vp, _vp, _vp, _vp]
        L.raftx_debug_math_table.restype = C.c_int
        L.raftx_build_designs.argtypes = [_vp, C.c_int, _vp, _vp, _vp, _vp, _vp, _vp, _vp, C.c_double, C.c_double, C.c_int,
                                          _vp, C.c_int, _vp, _vp, _vp, _vp, _vp, _vp]
        L.raftx_build_designs.restype = C.c_int
        L.raftx_fetch_strips.argtypes = [_vp, _vp, _vp]
        L.raftx_fetch_strips.restype = C.c_int
        L.raftx_fetch_statics.argtypes = [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]
        L.raftx_fetch_statics.restype = C.c_int
        L.raftx_last_kernel_ms.argtypes = [_vp]
        L.raftx_last_kernel_ms.restype = C.c_double
        L.raftx_sweep_stats.argtypes = [_vp, C.c_int, _vp, _vp, _vp, _vp, _vp, _vp, _vp, C.c_double, C.c_double, C.c_int,
                                        _vp, _vp, _vp, _vp, C.c_int, C.c_int, C.c_int, _vp, _vp, C.c_double, C.c_double,
                                        C.c_double, _vp, _vp, C.c_int, C.c_double, C.c_double, C.c_int, C.c_int,
                                        _vp, _vp, _vp, _vp, _vp, _vp]
        L.raftx_sweep_stats.restype = C.c_int
        L.raftx_sweep_submit.argtypes = [_vp, C.c_int, C.c_int, _vp, _vp, _vp, _vp, _vp, _vp, _vp, C.c_double, C.c_double, C.c_int,
                                         _vp, _vp, _vp, _vp, C.c_int, C.c_int, C.c_int, _vp, _vp, C.c_double, C.c_double, C.c_double,
                                         _vp, _vp, C.c_int, C.c_double, C.c_double, C.c_int, _vp, _vp, _vp, _vp, _vp]
        L.raftx_sweep_submit.restype = C.c_int
        L.raftx_sweep_prepare.argtypes = L.raftx_sweep_submit.argtypes
        L.raftx_sweep_prepare.restype = C.c_int
        L.raftx_strip_kinematics.argtypes = [_vp, C.c_int, C.c_int, _vp, _vp, _vp]
        L.raftx_strip_kinematics.restype = C.c_int
        L.raftx_strip_drag.argtypes = [_vp, C.c_int, C.c_int, _vp, C.c_int, _vp, _vp]
        L.raftx_strip_drag.restype = C.c_int
        L.raftx_variant_program.argtypes = [_vp, C.c_int, _vp, _vp, _vp, _vp, _vp, C.c_int, _vp, _vp, _vp, _vp, _vp]
        L.raftx_variant_program.restype = C.c_int
        L.raftx_expand_variants.argtypes = [_vp, C.c_int, _vp, _vp, _vp, _vp]
        L.raftx_expand_variants.restype = C.c_int
        L.raftx_sweep_prepare_variants.argtypes = [_vp, C.c_int, C.c_int, _vp, _vp, C.c_double, C.c_double, C.c_int,
                                                   _vp, _vp, _vp, _vp, C.c_int, C.c_int, C.c_int, _vp, _vp, C.c_double, C.c_double, C.c_double,
                                                   _vp, _vp, C.c_int, C.c_double, C.c_double, C.c_int, _vp, _vp, _vp, _vp, _vp]
        L.raftx_sweep_prepare_variants.restype = C.c_int
        L.raftx_sweep_wait.argtypes = [_vp, C.c_int, _vp]
        L.raftx_sweep_wait.restype = C.c_int
        L.raftx_response_stats.argtypes = [_vp, C.c_int, C.c_int, C.c_int, C.c_int, _vp, _vp, _vp, _vp, C.c_double, _vp, _vp]
        L.raftx_response_stats.restype = C.c_int
        L.raftx_sweep_solve_span.argtypes = [_vp, C.c_int, _vp, _vp]
        L.raftx_sweep_solve_span.restype = C.c_int
        L.raftx_sweep_generation.argtypes = [_vp, C.c_int, _vp, _vp]
        L.raftx_sweep_generation.restype = C.c_int
        L.raftx_sweep_cancel.argtypes = [_vp, C.c_int]
        L.raftx_sweep_cancel.restype = C.c_int
        L.raftx_device_count.argtypes = []
        L.raftx_device_count.restype = C.c_int
        L.raftx_comm_unique_id.argtypes = [_vp, _vp]
        L.raftx_comm_init.argtypes = [_vp, C.c_int, C.c_int, _vp]
        L.raftx_comm_destroy.argtypes = [_vp]
        L.raftx_comm_broadcast.argtypes = [_vp, _vp, C.c_size_t, C.c_int]
        L.raftx_comm_gather_rows.argtypes = [_vp, _vp, _vp, C.c_size_t, _vp, C.c_int]
        L.raftx_comm_gather_xi.argtypes = [_vp, _vp, _vp, C.c_int]
        L.raftx_comm_reduce_sum.argtypes = [_vp, _vp, C.c_size_t, C.c_int]
        for f in (L.raftx_comm_unique_id, L.raftx_comm_init, L.raftx_comm_destroy, L.raftx_comm_broadcast,
                  L.raftx_comm_gather_rows, L.raftx_comm_gather_xi, L.raftx_comm_reduce_sum):
            f.restype = C.c_int

    @property
    def version(self):
        return int(self.lib.raftx_version())

    @property
    def is_device(self):
        return bool(self.lib.raftx_is_device())

    def context(self, device_id=0):
        return Context(self, device_id)

    def device_count(self):
        """GPUs this process can open (raftx_device_count; 0 for the oracle or a host without a GPU)."""
        return int(self.lib.raftx_device_count())

    def device_locality(self, device_id=0):
        """(PCI address, NUMA node or -1) of a device: raftx_device_locality."""
        buf = C.create_string_buffer(64)
        node = C.c_int(-1)
        rc = self.lib.raftx_device_locality(int(device_id), buf, len(buf), C.byref(node))
        if rc != 0:
            raise RaftxError("raftx_device_locality(device=%d) failed (rc=%d)" % (device_id, rc))
        return buf.value.decode(), node.value


class Context:
    """One raftx_ctx: owns a stream and the device-resident design/case tables."""

    def __init__(self, rlib, device_id=0):
        self.rlib = rlib
        self._h = _vp()
        rc = rlib.lib.raftx_ctx_create(int(device_id), C.byref(self._h))
        if rc != 0 or not self._h:
            raise RaftxError("raftx_ctx_create(device=%d) failed (rc=%d) in %s"
                             % (device_id, rc, rlib.path))
        self.nDesign = self.nCase = self.nHead = self.nw = 0
        self._comm = None
        # bumped by every call that replaces what is resident on the context (designs, cases, sweep crossings): callers
        # that skip an upload because "their" tables are already there compare it (raft_amd/dropin.py Engine._upload)
        self.resident_generation = 0

    def close(self):
        if self._h and self.poisoned:     # a thread may still be inside the library on this context: leak it, never destroy it
            self._h = _vp()
            return
        if self._h:
            for ptr in list(getattr(self, "_pinned", {}).values()):       # page-locked buffers still out: release them
                self.rlib.lib.raftx_host_free(self._h, _vp(ptr))
            self._pinned = {}
            self.rlib.lib.raftx_ctx_destroy(self._h)
            self._h = _vp()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    poisoned = None                       # set by raft_amd/comm.py when a call on this context never returned

    def _check(self, rc, what):
        if self.poisoned:
            raise RaftxError("%s: this context must not be used any more: %s" % (what, self.poisoned))
        if rc != 0:
            msg = self.rlib.lib.raftx_last_error(self._h)
            raise RaftxError("%s failed (rc=%d): %s" % (what, rc, (msg or b"").decode()))

    # ------------------------------------------------------------- uploads
    def upload_designs(self, strip_tables, M0, B0, C0, nw, MBw=None):
        """strip_tables: list of raft_amd.strips.StripTable (one per design)."""
        self.resident_generation += 1
        nD = len(strip_tables)
        off = np.zeros(nD + 1, dtype=np.int64)
        cmoff = np.zeros(nD + 1, dtype=np.int64)
        for i, t in enumerate(strip_tables):
            off[i + 1] = off[i] + t.n
            cmoff[i + 1] = cmoff[i] + (0 if t.cm_mcf is None else t.cm_mcf.shape[0])
        strips = np.concatenate([t.strips for t in strip_tables], axis=0) if nD else np.zeros((0, NFIELD))
        strips = _f64(strips.reshape(-1, NFIELD))
        cm = None
        if cmoff[-1] > 0:
            cm = _c128(np.concatenate([t.cm_mcf for t in strip_tables if t.cm_mcf is not None], axis=0),
                       (cmoff[-1], 2, nw), "CmMCF")
        return self.upload_designs_raw(off, strips, M0, B0, C0, nw, MBw, cmoff if cm is not None else None, cm)

    def upload_designs_raw(self, off, strips, M0, B0, C0, nw, MBw=None, cmoff=None, cm=None):
        self.resident_generation += 1
        off = np.ascontiguousarray(off, dtype=np.int64)
        nD = len(off) - 1
        strips = _f64(strips)
        if strips.size != off[-1] * NFIELD:
            raise ValueError("strips has %d values, offsets imply %d" % (strips.size, off[-1] * NFIELD))
        M0 = _f64(M0, (nD, 6, 6), "M0")
        B0 = _f64(B0, (nD, 6, 6), "B0")
        C0 = _f64(C0, (nD, 6, 6), "C0")
        if MBw is not None:
            MBw = _f64(MBw, (nD, 2, 6, 6, nw), "MBw")
        if cm is not None:
            cmoff = np.ascontiguousarray(cmoff, dtype=np.int64)
            cm = _c128(cm, (cmoff[-1], 2, nw), "CmMCF")
        rc = self.rlib.lib.raftx_upload_designs(self._h, nD, _ptr(off), _ptr(strips), NFIELD,
                                                _ptr(M0), _ptr(B0), _ptr(C0), int(nw), _ptr(MBw),
                                                _ptr(cmoff) if cm is not None else None, _ptr(cm))
        self._check(rc, "raftx_upload_designs")
        self.nDesign = nD
        self._nw_designs = int(nw)

    def build_designs(self, member_off, members, station_off, stations, M0, B0, C0, nw, pose=None, rho=1025.0,
                      g=9.81, k=None, add_mask=0, MBw=None, cap_off=None, caps=None, Fz_moor=None):
        """Geometry -> resident strip tables (+ statics) on the device: raftx_build_designs.  Member / station / cap
        records as raft_amd/geometry.py packs them.  Returns the strip offsets [nDesign+1]."""
        self.resident_generation += 1
        member_off = np.ascontiguousarray(member_off, dtype=np.int64)
        station_off = np.ascontiguousarray(station_off, dtype=np.int64)
        nD = len(member_off) - 1
        members = _f64(members, (member_off[-1], 16), "members")
        if len(station_off) != member_off[-1] + 1:
            raise ValueError("station_off has %d entries, expected %d" % (len(station_off), member_off[-1] + 1))
        stations = _f64(stations, (station_off[-1], 16), "stations")
        M0 = _f64(M0, (nD, 6, 6), "M0")
        B0 = _f64(B0, (nD, 6, 6), "B0")
        C0 = _f64(C0, (nD, 6, 6), "C0")
        if pose is not None:
            pose = _f64(pose, (nD, 6), "pose")
        if k is not None:
            k = _f64(k, (nw,), "k")
        if MBw is not None:
            MBw = _f64(MBw, (nD, 2, 6, 6, nw), "MBw")
        if cap_off is not None:
            cap_off = np.ascontiguousarray(cap_off, dtype=np.int64)
            if len(cap_off) != member_off[-1] + 1:
                raise ValueError("cap_off has %d entries, expected %d" % (len(cap_off), member_off[-1] + 1))
            caps = _f64(caps, (cap_off[-1], 4), "caps")
            if caps.size == 0:
                caps = np.zeros((1, 4))
        if Fz_moor is not None:
            Fz_moor = _f64(Fz_moor, (nD,), "Fz_moor")
        off = np.zeros(nD + 1, dtype=np.int64)
        rc = self.rlib.lib.raftx_build_designs(self._h, nD, _ptr(member_off), _ptr(members), _ptr(station_off),
                                               _ptr(stations), _ptr(cap_off), _ptr(caps if cap_off is not None else None),
                                               _ptr(pose), float(rho), float(g), int(nw), _ptr(k),
                                               int(add_mask), _ptr(M0), _ptr(B0), _ptr(C0), _ptr(MBw), _ptr(Fz_moor), _ptr(off))
        self._check(rc, "raftx_build_designs")
        self.nDesign = nD
        self._nw_designs = int(nw)
        self._strip_off = off
        return off

    def _sweep_prepare(self, tables, M0, B0, C0, w, k, depth, zeta, beta, pose, Fz_moor, want_Xi, Xi_out):
        """Checked, contiguous views of the inputs of a sweep crossing + freshly allocated outputs."""
        member_off = np.ascontiguousarray(tables.member_off, dtype=np.int64)
        station_off = np.ascontiguousarray(tables.station_off, dtype=np.int64)
        nD = len(member_off) - 1
        members = _f64(tables.members, (member_off[-1], 16), "members")
        stations = _f64(tables.stations, (station_off[-1], 16), "stations")
        cap_off = caps = None
        if getattr(tables, "cap_off", None) is not None:
            cap_off = np.ascontiguousarray(tables.cap_off, dtype=np.int64)
            caps = _f64(tables.caps, (cap_off[-1], 4), "caps")
            if caps.size == 0:
                caps = np.zeros((1, 4))
        M0, B0, C0 = _f64(M0, (nD, 6, 6), "M0"), _f64(B0, (nD, 6, 6), "B0"), _f64(C0, (nD, 6, 6), "C0")
        pose = None if pose is None else _f64(pose, (nD, 6), "pose")
        Fz = None if Fz_moor is None else _f64(Fz_moor, (nD,), "Fz_moor")
        w = _f64(w)
        nw = len(w)
        k = _f64(k, (nw,), "k")
        zeta = _f64(zeta)
        if zeta.ndim == 2:
            zeta, beta = zeta[None], np.asarray(beta, dtype=np.float64)[None]
        nC, nH = zeta.shape[0], zeta.shape[1]
        zeta = _f64(zeta, (nC, nH, nw), "zeta")
        beta = _f64(beta, (nC, nH), "beta")
        Xi = Xi_out
        if Xi is None and want_Xi:
            Xi = np.empty((nD, nC, nH, 6, nw), dtype=np.complex128)
        if Xi is not None and (Xi.dtype != np.complex128 or Xi.shape != (nD, nC, nH, 6, nw) or not Xi.flags["C_CONTIGUOUS"]):
            raise ValueError("Xi_out must be a C-contiguous complex128 array of shape %s" % ((nD, nC, nH, 6, nw),))
        out = dict(std=np.empty((nD, nC, 6)), niter=np.zeros((nD, nC), dtype=np.int32), flags=np.zeros((nD, nC), dtype=np.int32),
                   Xi=Xi, strip_off=np.zeros(nD + 1, dtype=np.int64), timing_ms=np.zeros(4))
        inputs = (member_off, members, station_off, stations, cap_off, caps, pose, M0, B0, C0, Fz, w, k, zeta, beta)
        return nD, nC, nH, nw, inputs, out

    def sweep_prepare(self, slot, tables, M0, B0, C0, w, k, depth, zeta, beta, nIter, tol=0.01, XiStart=0.1, pose=None,
                      rho=1025.0, g=9.81, rho_wave=1025.0, g_wave=9.81, add_mask=7, Fz_moor=None, n_chunk=0, want_Xi=False,
                      Xi_out=None):
        """First stage of a sweep crossing on ``slot`` (0 .. 3): descriptor upload + member pass are enqueued
        (raftx_sweep_prepare); returns a handle for ``sweep_launch`` / ``sweep_wait``.  The handle keeps the input and
        output arrays alive; do not modify the inputs before ``sweep_wait``."""
        self.resident_generation += 1
        nD, nC, nH, nw, inputs, out = self._sweep_prepare(tables, M0, B0, C0, w, k, depth, zeta, beta, pose, Fz_moor, want_Xi, Xi_out)
        (member_off, members, station_off, stations, cap_off, caps, pose, M0, B0, C0, Fz, w, k, zeta, beta) = inputs
        rc = self.rlib.lib.raftx_sweep_prepare(self._h, int(slot), nD, _ptr(member_off), _ptr(members), _ptr(station_off), _ptr(stations),
                                               _ptr(cap_off), _ptr(caps), _ptr(pose), float(rho), float(g), int(add_mask),
                                               _ptr(M0), _ptr(B0), _ptr(C0), _ptr(Fz), nC, nH, nw, _ptr(w), _ptr(k), float(depth),
                                               float(rho_wave), float(g_wave), _ptr(zeta), _ptr(beta), int(nIter), float(tol),
                                               float(XiStart), int(n_chunk), _ptr(out["std"]), _ptr(out["niter"]), _ptr(out["flags"]),
                                               _ptr(out["Xi"]), _ptr(out["strip_off"]))
        self._check(rc, "raftx_sweep_prepare")
        return dict(slot=int(slot), inputs=inputs, out=out)

    # ------------------------------------------------------------- per-strip by-products on request
    def strip_kinematics(self, design, n_strips, icase=0):
        """(u, ud [nHead,S,3,nw], pDyn [nHead,S,nw]) of the S = n_strips strips of resident design ``design`` under the
        resident sea state ``icase`` (raftx_strip_kinematics): what Member.computeWaveKinematics keeps on the member."""
        S, nH, nw = int(n_strips), self.nHead, self.nw
        u = np.empty((nH, S, 3, nw), dtype=np.complex128)
        ud = np.empty((nH, S, 3, nw), dtype=np.complex128)
        p = np.empty((nH, S, nw), dtype=np.complex128)
        self._check(self.rlib.lib.raftx_strip_kinematics(self._h, int(design), int(icase), _ptr(u), _ptr(ud), _ptr(p)),
                    "raftx_strip_kinematics")
        return u, ud, p

    def strip_drag(self, design, n_strips, Xi, ih=0, icase=0):
        """(Bmat [S,3,3], F_exc_drag [S,3,nw]) of the linearisation about Xi [6,nw] (raftx_strip_drag): what
        Member.calcHydroLinearization / calcDragExcitation keep on the member."""
        S, nw = int(n_strips), self.nw
        Xi = _c128(Xi, (6, nw), "Xi")
        B = np.empty((S, 3, 3))
        F = np.empty((S, 3, nw), dtype=np.complex128)
        self._check(self.rlib.lib.raftx_strip_drag(self._h, int(design), int(icase), _ptr(Xi), int(ih), _ptr(B), _ptr(F)),
                    "raftx_strip_drag")
        return B, F

    # ------------------------------------------------------------- parametric variants of one base unit
    def variant_program(self, prog):
        """Installs a raft_amd.geometry.VariantProgram on the context (raftx_variant_program); None clears it."""
        if prog is None:
            self._check(self.rlib.lib.raftx_variant_program(self._h, 0, None, None, None, None, None, 0, None, None, None, None, None),
                        "raftx_variant_program")
            self._vprog = None
            return
        b = prog.base
        nM, nSt, nCap, nP = b.n, len(b.stations), len(b.caps), prog.n_param
        arrs = dict(members=_f64(b.members, (nM, 16), "members"), station_off=np.ascontiguousarray(b.station_off, dtype=np.int64),
                    stations=_f64(b.stations, (nSt, 16), "stations"), cap_off=np.ascontiguousarray(b.cap_off, dtype=np.int64),
                    caps=_f64(b.caps if nCap else np.zeros((1, 4)), (max(nCap, 1), 4), "caps"),
                    end_coef=_f64(prog.end_coef, (nM, 6, nP + 1), "end_coef"), end_edit=np.ascontiguousarray(prog.end_edit, dtype=np.int32),
                    head_cs=_f64(prog.head_cs, (nM, 2), "head_cs"), dia_coef=_f64(prog.dia_coef, (nSt, 2, nP + 1), "dia_coef"),
                    dia_edit=np.ascontiguousarray(prog.dia_edit, dtype=np.int32))
        rc = self.rlib.lib.raftx_variant_program(self._h, nM, _ptr(arrs["members"]), _ptr(arrs["station_off"]), _ptr(arrs["stations"]),
                                                 _ptr(arrs["cap_off"]), _ptr(arrs["caps"]), nP, _ptr(arrs["end_coef"]),
                                                 _ptr(arrs["end_edit"]), _ptr(arrs["head_cs"]), _ptr(arrs["dia_coef"]), _ptr(arrs["dia_edit"]))
        self._check(rc, "raftx_variant_program")
        self._vprog = (nM, nSt, nCap, nP)

    def expand_variants(self, params):
        """(members [nD*nM,16], stations [nD*nSt,16], caps [nD*nCap,4]) of the variants ``params`` [nD,nParam] of the installed
        program, written by the library (raftx_expand_variants): for checks -- the sweep path never downloads them."""
        if getattr(self, "_vprog", None) is None:
            raise RaftxError("expand_variants: no program installed (variant_program first)")
        nM, nSt, nCap, nP = self._vprog
        params = _f64(params)
        nD = params.shape[0]
        params = _f64(params, (nD, nP), "params")
        gm, gs, gc = np.empty((nD * nM, 16)), np.empty((nD * nSt, 16)), np.empty((nD * nCap, 4))
        self._check(self.rlib.lib.raftx_expand_variants(self._h, nD, _ptr(params), _ptr(gm), _ptr(gs), _ptr(gc) if nCap else None),
                    "raftx_expand_variants")
        return gm, gs, gc

    def sweep_prepare_variants(self, slot, params, M0, B0, C0, w, k, depth, zeta, beta, nIter, tol=0.01, XiStart=0.1, pose=None,
                               rho=1025.0, g=9.81, rho_wave=1025.0, g_wave=9.81, add_mask=7, Fz_moor=None, n_chunk=0, want_Xi=False,
                               Xi_out=None):
        """``sweep_prepare`` for variants of the installed program: ``params`` [nD,nParam] cross the bus, the descriptors are
        written on the device (raftx_sweep_prepare_variants).  sweep_launch / sweep_wait / sweep_cancel as usual."""
        if getattr(self, "_vprog", None) is None:
            raise RaftxError("sweep_prepare_variants: no program installed (variant_program first)")
        self.resident_generation += 1
        nP = self._vprog[3]
        params = _f64(params)
        nD = params.shape[0]
        params = _f64(params, (nD, nP), "params")
        M0, B0, C0 = _f64(M0, (nD, 6, 6), "M0"), _f64(B0, (nD, 6, 6), "B0"), _f64(C0, (nD, 6, 6), "C0")
        pose = None if pose is None else _f64(pose, (nD, 6), "pose")
        Fz = None if Fz_moor is None else _f64(Fz_moor, (nD,), "Fz_moor")
        w = _f64(w)
        nw = len(w)
        k = _f64(k, (nw,), "k")
        zeta = _f64(zeta)
        if zeta.ndim == 2:
            zeta, beta = zeta[None], np.asarray(beta, dtype=np.float64)[None]
        nC, nH = zeta.shape[0], zeta.shape[1]
        zeta, beta = _f64(zeta, (nC, nH, nw), "zeta"), _f64(beta, (nC, nH), "beta")
        Xi = Xi_out
        if Xi is None and want_Xi:
            Xi = np.empty((nD, nC, nH, 6, nw), dtype=np.complex128)
        if Xi is not None and (Xi.dtype != np.complex128 or Xi.shape != (nD, nC, nH, 6, nw) or not Xi.flags["C_CONTIGUOUS"]):
            raise ValueError("Xi_out must be a C-contiguous complex128 array of shape %s" % ((nD, nC, nH, 6, nw),))
        out = dict(std=np.empty((nD, nC, 6)), niter=np.zeros((nD, nC), dtype=np.int32), flags=np.zeros((nD, nC), dtype=np.int32),
                   Xi=Xi, strip_off=np.zeros(nD + 1, dtype=np.int64), timing_ms=np.zeros(4))
        inputs = (params, pose, M0, B0, C0, Fz, w, k, zeta, beta)
        rc = self.rlib.lib.raftx_sweep_prepare_variants(self._h, int(slot), nD, _ptr(params), _ptr(pose), float(rho), float(g), int(add_mask),
                                                        _ptr(M0), _ptr(B0), _ptr(C0), _ptr(Fz), nC, nH, nw, _ptr(w), _ptr(k), float(depth),
                                                        float(rho_wave), float(g_wave), _ptr(zeta), _ptr(beta), int(nIter), float(tol),
                                                        float(XiStart), int(n_chunk), _ptr(out["std"]), _ptr(out["niter"]),
                                                        _ptr(out["flags"]), _ptr(out["Xi"]), _ptr(out["strip_off"]))
        self._check(rc, "raftx_sweep_prepare_variants")
        return dict(slot=int(slot), inputs=inputs, out=out)

    def sweep_launch(self, handle):
        """Second stage: generation, fused fixed point and statistics of a prepared crossing are enqueued (raftx_sweep_launch)."""
        self._check(self.rlib.lib.raftx_sweep_launch(self._h, int(handle["slot"])), "raftx_sweep_launch")
        return handle

    def sweep_submit(self, slot, *args, **kw):
        """prepare + launch in one call (raftx_sweep_submit's two-stage form): returns the handle for ``sweep_wait``."""
        return self.sweep_launch(self.sweep_prepare(slot, *args, **kw))

    def sweep_wait(self, handle):
        """Block until the crossing of ``handle`` (from ``sweep_submit``) has finished; returns its results
        (dict as ``sweep_stats``)."""
        out = handle["out"]
        self._check(self.rlib.lib.raftx_sweep_wait(self._h, int(handle["slot"]), _ptr(out["timing_ms"])), "raftx_sweep_wait")
        span = np.zeros(2)                                # where the crossing's fused kernels ran on the device's clock (ms)
        self._check(self.rlib.lib.raftx_sweep_solve_span(self._h, int(handle["slot"]), _ptr(span[0:1]), _ptr(span[1:2])),
                    "raftx_sweep_solve_span")
        out["solve_span_ms"] = span
        nf, nb = C.c_int(0), C.c_int(0)                   # blocks whose tables the fused kernel built itself / all blocks
        self._check(self.rlib.lib.raftx_sweep_generation(self._h, int(handle["slot"]), C.byref(nf), C.byref(nb)),
                    "raftx_sweep_generation")
        out["generation_fused_blocks"] = (nf.value, nb.value)
        handle["inputs"] = None
        return out

    def sweep_cancel(self, handle):
        """Retire a crossing that was prepared and will not be launched (raftx_sweep_cancel)."""
        self._check(self.rlib.lib.raftx_sweep_cancel(self._h, int(handle["slot"])), "raftx_sweep_cancel")
        handle["inputs"] = None

    def sweep_stats(self, tables, M0, B0, C0, w, k, depth, zeta, beta, nIter, tol=0.01, XiStart=0.1, pose=None,
                    rho=1025.0, g=9.81, rho_wave=1025.0, g_wave=9.81, add_mask=7, Fz_moor=None, n_chunk=0, n_worker=0,
                    want_Xi=False, Xi_out=None):
        """One whole sweep crossing in ONE library call (raftx_sweep_stats): member descriptions (``tables``: a
        raft_amd.geometry.DesignTables) in, motion statistics + iteration counts (+ responses) out; inside the library the
        designs are cut into blocks whose descriptor upload, table generation, fixed-point kernel and downloads are
        pipelined over internal streams.  Returns dict(std [nD,nC,6], niter, flags [nD,nC], Xi or None, strip_off [nD+1],
        timing_ms [wall, generation kernels, solve kernels, statistics kernels]).  Unlike build_designs + solve, nothing
        stays resident on this context afterwards."""
        self.resident_generation += 1
        nD, nC, nH, nw, inputs, out = self._sweep_prepare(tables, M0, B0, C0, w, k, depth, zeta, beta, pose, Fz_moor, want_Xi, Xi_out)
        (member_off, members, station_off, stations, cap_off, caps, pose, M0, B0, C0, Fz, w, k, zeta, beta) = inputs
        rc = self.rlib.lib.raftx_sweep_stats(self._h, nD, _ptr(member_off), _ptr(members), _ptr(station_off), _ptr(stations),
                                             _ptr(cap_off), _ptr(caps), _ptr(pose), float(rho), float(g), int(add_mask),
                                             _ptr(M0), _ptr(B0), _ptr(C0), _ptr(Fz), nC, nH, nw, _ptr(w), _ptr(k), float(depth),
                                             float(rho_wave), float(g_wave), _ptr(zeta), _ptr(beta), int(nIter), float(tol),
                                             float(XiStart), int(n_chunk), int(n_worker), _ptr(out["std"]), _ptr(out["niter"]),
                                             _ptr(out["flags"]), _ptr(out["Xi"]), _ptr(out["strip_off"]), _ptr(out["timing_ms"]))
        self._check(rc, "raftx_sweep_stats")
        return out

    def fetch_strips(self, n_strips, n_cm_rows=0):
        """(strips [n,32], cm [rows,2,nw] or None) generated by the last build_designs."""
        strips = np.empty((int(n_strips), NFIELD))
        cm = np.empty((int(n_cm_rows), 2, self._nw_designs), dtype=np.complex128) if n_cm_rows else None
        self._check(self.rlib.lib.raftx_fetch_strips(self._h, _ptr(strips), _ptr(cm)), "raftx_fetch_strips")
        return strips, cm

    def fetch_statics(self):
        """dict(A_morison, C_hydro, M_struc, C_struc [nD,6,6]; W_hydro, W_struc [nD,6]; props [nD,12])."""
        nD = self.nDesign
        out = dict(A_morison=np.empty((nD, 6, 6)), C_hydro=np.empty((nD, 6, 6)), W_hydro=np.empty((nD, 6)),
                   M_struc=np.empty((nD, 6, 6)), C_struc=np.empty((nD, 6, 6)), W_struc=np.empty((nD, 6)),
                   props=np.empty((nD, 12)))
        rc = self.rlib.lib.raftx_fetch_statics(self._h, _ptr(out["A_morison"]), _ptr(out["C_hydro"]),
                                               _ptr(out["W_hydro"]), _ptr(out["M_struc"]), _ptr(out["C_struc"]),
                                               _ptr(out["W_struc"]), _ptr(out["props"]))
        self._check(rc, "raftx_fetch_statics")
        return out

    def upload_cases(self, w, k, depth, rho, g, zeta, beta):
        self.resident_generation += 1
        w = _f64(w)
        nw = w.shape[0]
        k = _f64(k, (nw,), "k")
        zeta = _f64(zeta)
        if zeta.ndim != 3 or zeta.shape[2] != nw:
            raise ValueError("zeta must be [nCase,nHead,nw]")
        nC, nH = zeta.shape[0], zeta.shape[1]
        beta = _f64(beta, (nC, nH), "beta")
        rc = self.rlib.lib.raftx_upload_cases(self._h, nC, nH, nw, _ptr(w), _ptr(k),
                                              float(depth), float(rho), float(g), _ptr(zeta), _ptr(beta))
        self._check(rc, "raftx_upload_cases")
        self.nCase, self.nHead, self.nw = nC, nH, nw

    # ------------------------------------------------------------- compute
    def excitation(self, out=None):
        """F_iner [nDesign,nCase,nHead,6,nw]; out: a preallocated C-contiguous complex128 array of that shape (e.g. page-locked)"""
        shape = (self.nDesign, self.nCase, self.nHead, 6, self.nw)
        F = np.empty(shape, dtype=np.complex128) if out is None else _c128(out, shape, "out")
        if out is not None and F is not out:
            raise ValueError("excitation: out must be a C-contiguous complex128 array")
        self._check(self.rlib.lib.raftx_excitation(self._h, _ptr(F)), "raftx_excitation")
        return F

    def linearize(self, Xi, want_F=True):
        Xi = _c128(Xi, (self.nDesign, self.nCase, 6, self.nw), "Xi")
        B = np.empty((self.nDesign, self.nCase, 6, 6), dtype=np.float64)
        F = np.empty((self.nDesign, self.nCase, self.nHead, 6, self.nw), dtype=np.complex128) if want_F else None
        self._check(self.rlib.lib.raftx_linearize(self._h, _ptr(Xi), _ptr(B), _ptr(F)), "raftx_linearize")
        return B, F

    def solve_dynamics(self, nIter, tol=0.01, XiStart=0.1, F_extra=None,
                       want_Xi=True, want_B=False, want_F=False, want_Z=False):
        nD, nC, nH, nw = self.nDesign, self.nCase, self.nHead, self.nw
        if F_extra is not None:
            F_extra = _c128(F_extra, (nD, nC, nH, 6, nw), "F_extra")
        out = {}
        Xi = np.empty((nD, nC, nH, 6, nw), dtype=np.complex128) if want_Xi else None
        niter = np.zeros((nD, nC), dtype=np.int32)
        flags = np.zeros((nD, nC), dtype=np.int32)
        B = np.empty((nD, nC, 6, 6), dtype=np.float64) if want_B else None
        F = np.empty((nD, nC, nH, 6, nw), dtype=np.complex128) if want_F else None
        Z = np.empty((nD, nC, 6, 6, nw), dtype=np.complex128) if want_Z else None
        rc = self.rlib.lib.raftx_solve_dynamics(self._h, int(nIter), float(tol), float(XiStart),
                                                _ptr(F_extra), _ptr(Xi), _ptr(niter), _ptr(flags),
                                                _ptr(B), _ptr(F), _ptr(Z))
        self._check(rc, "raftx_solve_dynamics")
        out.update(Xi=Xi, niter=niter, flags=flags, B_drag=B, F_wave=F, Z=Z)
        return out

    def solve_dynamics_device(self, nIter, tol=0.01, XiStart=0.1, F_extra=None, want_mask=0):
        """Launch only; results stay in HBM (see fetch_results)."""
        if F_extra is not None:
            F_extra = _c128(F_extra, (self.nDesign, self.nCase, self.nHead, 6, self.nw), "F_extra")
        rc = self.rlib.lib.raftx_solve_dynamics_device(self._h, int(nIter), float(tol), float(XiStart),
                                                       _ptr(F_extra), int(want_mask))
        self._check(rc, "raftx_solve_dynamics_device")

    def pinned_empty(self, shape, dtype=np.complex128):
        """An uninitialised array in page-locked host memory (raftx_host_alloc): D2H copies into it run at full PCIe
        rate.  Release it with ``free_pinned(array)`` (before the context is closed); do not keep views past that."""
        dtype = np.dtype(dtype)
        n = int(np.prod(shape)) * dtype.itemsize
        ptr = _vp()
        self._check(self.rlib.lib.raftx_host_alloc(self._h, n, C.byref(ptr)), "raftx_host_alloc")
        buf = (C.c_char * max(n, 1)).from_address(ptr.value)
        arr = np.frombuffer(buf, dtype=dtype, count=int(np.prod(shape))).reshape(shape)
        self._pinned = getattr(self, "_pinned", {})
        self._pinned[arr.__array_interface__["data"][0]] = ptr.value
        return arr

    def free_pinned(self, arr):
        addr = arr.__array_interface__["data"][0]
        ptr = getattr(self, "_pinned", {}).pop(addr, None)
        if ptr is None:
            raise ValueError("not an array of pinned_empty of this context")
        self._check(self.rlib.lib.raftx_host_free(self._h, _vp(ptr)), "raftx_host_free")

    def fetch_results(self, want_Xi=True, want_B=False, want_F=False, want_Z=False, Xi_out=None):
        """Xi_out: optional preallocated C-contiguous complex128 [nDesign,nCase,nHead,6,nw] buffer (e.g. a slice of a
        larger array along the design axis) the responses are copied into instead of a fresh array."""
        nD, nC, nH, nw = self.nDesign, self.nCase, self.nHead, self.nw
        if Xi_out is not None:
            if Xi_out.dtype != np.complex128 or Xi_out.shape != (nD, nC, nH, 6, nw) or not Xi_out.flags["C_CONTIGUOUS"]:
                raise ValueError("Xi_out must be a C-contiguous complex128 array of shape %s" % ((nD, nC, nH, 6, nw),))
            Xi = Xi_out
        else:
            Xi = np.empty((nD, nC, nH, 6, nw), dtype=np.complex128) if want_Xi else None
        niter = np.zeros((nD, nC), dtype=np.int32)
        flags = np.zeros((nD, nC), dtype=np.int32)
        B = np.empty((nD, nC, 6, 6), dtype=np.float64) if want_B else None
        F = np.empty((nD, nC, nH, 6, nw), dtype=np.complex128) if want_F else None
        Z = np.empty((nD, nC, 6, 6, nw), dtype=np.complex128) if want_Z else None
        rc = self.rlib.lib.raftx_fetch_results(self._h, _ptr(Xi), _ptr(niter), _ptr(flags), _ptr(B), _ptr(F), _ptr(Z))
        self._check(rc, "raftx_fetch_results")
        return dict(Xi=Xi, niter=niter, flags=flags, B_drag=B, F_wave=F, Z=Z)

    def solve_system_resident(self, nUnit, Mc=None, Bc=None, Cc=None):
        """Coupled array response from the resident Z / F_wave (solve_dynamics_device with WANT_Z|WANT_FWAVE)."""
        nG, n = self.nDesign // nUnit, 6 * nUnit
        Mc = None if Mc is None else _f64(Mc, (nG, n, n), "Mc")
        Bc = None if Bc is None else _f64(Bc, (nG, n, n), "Bc")
        Cc = None if Cc is None else _f64(Cc, (nG, n, n), "Cc")
        Xi = np.empty((nG, self.nCase, self.nHead, n, self.nw), dtype=np.complex128)
        rc = self.rlib.lib.raftx_solve_system_resident(self._h, int(nUnit), _ptr(Mc), _ptr(Bc), _ptr(Cc), _ptr(Xi))
        self._check(rc, "raftx_solve_system_resident")
        return Xi

    def qtf_force(self, w2, w, dw, S0, qtf=None, n_set=None):
        """(f_mean [nSet,6], f [nSet,6,nw]) from QTFs: qtf [nSet,nw2,nw2,6], or None to use the QTFs left resident by
        the last qtf_slender call (give n_set)."""
        w2, w = _f64(w2), _f64(w)
        S0 = _f64(S0)
        nS = S0.shape[0] if n_set is None else int(n_set)
        S0 = _f64(S0, (nS, len(w)), "S0")
        if qtf is not None:
            qtf = _c128(qtf, (nS, len(w2), len(w2), 6), "qtf")
        f_mean = np.empty((nS, 6))
        f = np.empty((nS, 6, len(w)))
        rc = self.rlib.lib.raftx_qtf_force(self._h, nS, len(w2), _ptr(w2), _ptr(qtf), len(w), _ptr(w), float(dw), _ptr(S0),
                                           _ptr(f_mean), _ptr(f))
        self._check(rc, "raftx_qtf_force")
        return f_mean, f

    def qtf_kay(self, tables, beta, w2, k2, depth, rho, g, Nm=10, fetch=False):
        """Kim & Yue correction tables of the sets' MacCamy-Fuchs members on the device (raftx_qtf_kay); the result is
        consumed by the next qtf_slender call that passes kay=None.  Returns the table [nSet,nw2,nw2,6] if fetch."""
        from .qtf import kay_items, QK_N
        nS = len(tables)
        w2 = _f64(w2)
        nw2 = len(w2)
        k2 = _f64(k2, (nw2,), "k2")
        beta = _f64(beta, (nS,), "beta")
        items = [kay_items(t.kay_geom, float(b)) for t, b in zip(tables, beta)]
        ioff = np.concatenate([[0], np.cumsum([len(i) for i in items])]).astype(np.int64)
        flat = _f64(np.concatenate(items, axis=0)) if nS else np.zeros((0, QK_N))
        if flat.size == 0:
            flat = np.zeros((1, QK_N))
        out = np.empty((nS, nw2, nw2, 6), dtype=np.complex128) if fetch else None
        rc = self.rlib.lib.raftx_qtf_kay(self._h, nS, nw2, _ptr(w2), _ptr(k2), float(depth), float(rho), float(g),
                                         _ptr(ioff), _ptr(flat), _ptr(beta), int(Nm), _ptr(out))
        self._check(rc, "raftx_qtf_kay")
        return out

    def qtf_slender(self, tables, Xi, beta, w2, k2, depth, rho, g, Mstruc, kay=None, fetch=True, rows=None):
        """Batch of slender-body QTFs: tables = list of raft_amd.qtf.QtfTable (one per set), Xi [nSet,6,nw2],
        beta [nSet], Mstruc [nSet,6,6], kay [nSet,nw2,nw2,6] or None -> qtf [nSet,nw2,nw2,6].
        rows=(offset, stride): only the rows w1 = w2[offset::stride] and their mirrors (zeros elsewhere), the
        interleaved partition of ONE QTF over ranks (raftx_qtf_slender_rows)."""
        nS = len(tables)
        w2 = _f64(w2)
        nw2 = len(w2)
        k2 = _f64(k2, (nw2,), "k2")
        soff = np.zeros(nS + 1, dtype=np.int64)
        moff = np.zeros(nS + 1, dtype=np.int64)
        for i, t in enumerate(tables):
            soff[i + 1] = soff[i] + t.strips.shape[0]
            moff[i + 1] = moff[i] + t.members.shape[0]
        strips = _f64(np.concatenate([t.strips for t in tables], axis=0)) if nS else np.zeros((0, 24))
        members = _f64(np.concatenate([t.members for t in tables], axis=0)) if nS else np.zeros((0, 16))
        Xi = None if Xi is None else _c128(Xi, (nS, 6, nw2), "Xi")      # None: RAOs of the resident responses (device)
        beta = _f64(beta, (nS,), "beta")
        Mstruc = _f64(Mstruc, (nS, 6, 6), "Mstruc")
        kay = None if kay is None else _c128(kay, (nS, nw2, nw2, 6), "kay")
        qtf = np.empty((nS, nw2, nw2, 6), dtype=np.complex128) if fetch else None
        if rows is not None:
            rc = self.rlib.lib.raftx_qtf_slender_rows(self._h, nS, nw2, _ptr(w2), _ptr(k2), float(depth), float(rho), float(g),
                                                      _ptr(soff), _ptr(strips), _ptr(moff), _ptr(members), _ptr(Xi),
                                                      _ptr(beta), _ptr(Mstruc), _ptr(kay), int(rows[0]), int(rows[1]), _ptr(qtf))
            self._check(rc, "raftx_qtf_slender_rows")
            return qtf
        rc = self.rlib.lib.raftx_qtf_slender(self._h, nS, nw2, _ptr(w2), _ptr(k2), float(depth), float(rho), float(g),
                                             _ptr(soff), _ptr(strips), _ptr(moff), _ptr(members), _ptr(Xi), _ptr(beta),
                                             _ptr(Mstruc), _ptr(kay), _ptr(qtf))
        self._check(rc, "raftx_qtf_slender")
        return qtf

    def set_linearisation_point(self, XiLast0=None, keep_last=True):
        """Next solve starts from XiLast0 [nDesign,nCase,6,nw] (one-shot); keep_last: solves export their last
        linearisation point (fetch_linearisation_point)."""
        if XiLast0 is not None:
            XiLast0 = _c128(XiLast0, (self.nDesign, self.nCase, 6, self.nw), "XiLast0")
        rc = self.rlib.lib.raftx_set_linearisation_point(self._h, _ptr(XiLast0), 1 if keep_last else 0)
        self._check(rc, "raftx_set_linearisation_point")

    def fetch_linearisation_point(self):
        X = np.empty((self.nDesign, self.nCase, 6, self.nw), dtype=np.complex128)
        self._check(self.rlib.lib.raftx_fetch_linearisation_point(self._h, _ptr(X)), "raftx_fetch_linearisation_point")
        return X

    def channel_stats(self, L, pow, dw, want_psd=False):
        """std [nDesign,nCase,nChan] (and PSD) of the linear channels y_c = w^pow[c] * L[d,c,:] . Xi of the resident results."""
        L = _f64(L)
        if L.ndim == 2:
            L = np.ascontiguousarray(np.broadcast_to(L, (self.nDesign,) + L.shape))
        nCh = L.shape[1]
        L = _f64(L, (self.nDesign, nCh, 6), "L")
        pw = np.ascontiguousarray(pow, dtype=np.int32)
        if pw.shape != (nCh,):
            raise ValueError("pow must have one entry per channel")
        std = np.empty((self.nDesign, self.nCase, nCh), dtype=np.float64)
        psd = np.empty((self.nDesign, self.nCase, nCh, self.nw), dtype=np.float64) if want_psd else None
        rc = self.rlib.lib.raftx_channel_stats(self._h, nCh, _ptr(L), _ptr(pw), float(dw), _ptr(std), _ptr(psd))
        self._check(rc, "raftx_channel_stats")
        return std, psd

    def bem_excitation(self, headings_deg, X_BEM, heading_adjust=None, xy_ref=None, F_add=None, fetch=False):
        """Potential-flow excitation with heading interpolation for every (design, case, heading) of the uploaded
        designs / sea states (raftx_bem_excitation); stays resident as the F_extra of the following solves.
        X_BEM [nDesign,nHeadBEM,6,nw] (wave-heading frame, per unit amplitude)."""
        heads = _f64(headings_deg)
        nHB = len(heads)
        X = _c128(X_BEM, (self.nDesign, nHB, 6, self.nw), "X_BEM")
        ha = None if heading_adjust is None else _f64(heading_adjust, (self.nDesign,), "heading_adjust")
        xy = None if xy_ref is None else _f64(xy_ref, (self.nDesign, 2), "xy_ref")
        shape = (self.nDesign, self.nCase, self.nHead, 6, self.nw)
        Fa = None if F_add is None else _c128(F_add, shape, "F_add")
        out = np.empty(shape, dtype=np.complex128) if fetch else None
        rc = self.rlib.lib.raftx_bem_excitation(self._h, nHB, _ptr(heads), _ptr(X), _ptr(ha), _ptr(xy), _ptr(Fa), _ptr(out))
        self._check(rc, "raftx_bem_excitation")
        return out

    def channel_stats_poly(self, L, dw, Gw=None, want_psd=False):
        """std [nDesign,nCase,nChan] (and PSD) of y_c = sum_p (i w)^p L[d,c,p,:] . Xi + Gw[d,c,:,w] . Xi of the resident
        results (raftx_channel_stats_poly).  L [nChan,3,6] or [nDesign,nChan,3,6]; Gw [..,nChan,6,nw] complex or None."""
        L = _f64(L)
        if L.ndim == 3:
            L = np.ascontiguousarray(np.broadcast_to(L, (self.nDesign,) + L.shape))
        nCh = L.shape[1]
        L = _f64(L, (self.nDesign, nCh, 3, 6), "L")
        if Gw is not None:
            Gw = np.asarray(Gw, dtype=np.complex128)
            if Gw.ndim == 3:
                Gw = np.broadcast_to(Gw, (self.nDesign,) + Gw.shape)
            Gw = _c128(np.ascontiguousarray(Gw), (self.nDesign, nCh, 6, self.nw), "Gw")
        std = np.empty((self.nDesign, self.nCase, nCh), dtype=np.float64)
        psd = np.empty((self.nDesign, self.nCase, nCh, self.nw), dtype=np.float64) if want_psd else None
        rc = self.rlib.lib.raftx_channel_stats_poly(self._h, nCh, _ptr(L), _ptr(Gw), float(dw), _ptr(std), _ptr(psd))
        self._check(rc, "raftx_channel_stats_poly")
        return std, psd

    def response_stats(self, w, L, Xi, dw, Gw=None, want_psd=False):
        """std [nChan] (and PSD [nChan,nw]) of y_c = sum_p (i w)^p L[c,p,:] . Xi + Gw[c,:,w] . Xi for a response the caller
        holds (raftx_response_stats): Xi [nResp,nDof,nw] with any number of DOFs, L [nChan,3,nDof], Gw [nChan,nDof,nw]."""
        Xi = _c128(np.ascontiguousarray(Xi), None, "Xi")
        nResp, nDof, nw = Xi.shape
        L = _f64(L)
        nCh = L.shape[0]
        L = _f64(L, (nCh, 3, nDof), "L")
        w = _f64(w, (nw,), "w")
        if Gw is not None:
            Gw = _c128(np.ascontiguousarray(Gw), (nCh, nDof, nw), "Gw")
        std = np.empty(nCh, dtype=np.float64)
        psd = np.empty((nCh, nw), dtype=np.float64) if want_psd else None
        rc = self.rlib.lib.raftx_response_stats(self._h, nCh, nDof, nResp, nw, _ptr(w), _ptr(L), _ptr(Gw), _ptr(Xi), float(dw),
                                                _ptr(std), _ptr(psd))
        self._check(rc, "raftx_response_stats")
        return std, psd

    def motion_stats(self, dw, want_psd=False):
        """std [nDesign,nCase,6] (rotations in deg) and optionally PSD [nDesign,nCase,6,nw] of the resident results."""
        std = np.empty((self.nDesign, self.nCase, 6), dtype=np.float64)
        psd = np.empty((self.nDesign, self.nCase, 6, self.nw), dtype=np.float64) if want_psd else None
        self._check(self.rlib.lib.raftx_motion_stats(self._h, float(dw), _ptr(std), _ptr(psd)), "raftx_motion_stats")
        return std, psd

    def solve_system(self, w, Zblk, F, Mc=None, Bc=None, Cc=None):
        Zblk = _c128(Zblk)
        nS, nU = Zblk.shape[0], Zblk.shape[1]
        nw = Zblk.shape[-1]
        n = 6 * nU
        F = _c128(F)
        nR = F.shape[1]
        if F.shape != (nS, nR, n, nw):
            raise ValueError("F must be [nSys,nRhs,6*nUnit,nw]")
        w = _f64(w, (nw,), "w")
        Mc = None if Mc is None else _f64(Mc, (nS, n, n), "Mc")
        Bc = None if Bc is None else _f64(Bc, (nS, n, n), "Bc")
        Cc = None if Cc is None else _f64(Cc, (nS, n, n), "Cc")
        Xi = np.empty((nS, nR, n, nw), dtype=np.complex128)
        rc = self.rlib.lib.raftx_solve_system(self._h, nS, nU, nR, nw, _ptr(w), _ptr(Zblk),
                                              _ptr(Mc), _ptr(Bc), _ptr(Cc), _ptr(F), _ptr(Xi))
        self._check(rc, "raftx_solve_system")
        return Xi

    def solve_dense(self, w, M, B, C_, F, want_Z=False):
        """Xi [nRhs,n,nw] (and Z [n,n,nw]) of one n-DOF unit: raftx_solve_dense.  M, B: [n,n] or [n,n,nw]."""
        w = _f64(w)
        nw = len(w)
        F = _c128(F)
        nR, n = F.shape[0], F.shape[1]
        if F.shape != (nR, n, nw):
            raise ValueError("F must be [nRhs,n,nw]")
        M, B = _f64(M), _f64(B)
        mask = 0
        for bit, A, name in ((1, M, "M"), (2, B, "B")):
            if A.shape == (n, n, nw):
                mask |= bit
            elif A.shape != (n, n):
                raise ValueError("%s must be [n,n] or [n,n,nw]" % name)
        C_ = _f64(C_, (n, n), "C")
        Xi = np.empty((nR, n, nw), dtype=np.complex128)
        Z = np.empty((n, n, nw), dtype=np.complex128) if want_Z else None
        rc = self.rlib.lib.raftx_solve_dense(self._h, n, nR, nw, _ptr(w), _ptr(M), _ptr(B), _ptr(C_), mask, _ptr(F),
                                             _ptr(Xi), _ptr(Z))
        self._check(rc, "raftx_solve_dense")
        return (Xi, Z) if want_Z else Xi

    def solve_dense_batch(self, w, M, B, C_, F, want_Z=False):
        """Xi [nSys,nRhs,n,nw] (and Z [nSys,n,n,nw]) of nSys n-DOF systems in one launch: raftx_solve_dense_batch.
        M, B: [nSys,n,n] or [nSys,n,n,nw]; C_ [nSys,n,n]; F [nSys,nRhs,n,nw]."""
        w = _f64(w)
        nw = len(w)
        F = _c128(F)
        nS, nR, n = F.shape[0], F.shape[1], F.shape[2]
        if F.shape != (nS, nR, n, nw):
            raise ValueError("F must be [nSys,nRhs,n,nw]")
        M, B = _f64(M), _f64(B)
        mask = 0
        for bit, A, name in ((1, M, "M"), (2, B, "B")):
            if A.shape == (nS, n, n, nw):
                mask |= bit
            elif A.shape != (nS, n, n):
                raise ValueError("%s must be [nSys,n,n] or [nSys,n,n,nw]" % name)
        C_ = _f64(C_, (nS, n, n), "C")
        Xi = np.empty((nS, nR, n, nw), dtype=np.complex128)
        Z = np.empty((nS, n, n, nw), dtype=np.complex128) if want_Z else None
        rc = self.rlib.lib.raftx_solve_dense_batch(self._h, nS, n, nR, nw, _ptr(w), _ptr(M), _ptr(B), _ptr(C_), mask, _ptr(F),
                                                   _ptr(Xi), _ptr(Z))
        self._check(rc, "raftx_solve_dense_batch")
        return (Xi, Z) if want_Z else Xi

    def dense_resident(self, w, M, B, C_):
        """Keeps M, B [nSet,n,n] or [nSet,n,n,nw] and C_ [nSet,n,n] of nSet units on the device for solve_dense_resident
        (raftx_dense_resident); dense_resident(None, None, None, None) releases them."""
        if w is None:
            self._check(self.rlib.lib.raftx_dense_resident(self._h, 0, 0, 0, None, None, None, None, 0), "raftx_dense_resident")
            self._dense = None
            return
        w = _f64(w)
        nw = len(w)
        M, B = _f64(M), _f64(B)
        nS, n = M.shape[0], M.shape[1]
        mask = 0
        for bit, A, name in ((1, M, "M"), (2, B, "B")):
            if A.shape == (nS, n, n, nw):
                mask |= bit
            elif A.shape != (nS, n, n):
                raise ValueError("%s must be [nSet,n,n] or [nSet,n,n,nw]" % name)
        C_ = _f64(C_, (nS, n, n), "C")
        rc = self.rlib.lib.raftx_dense_resident(self._h, nS, n, nw, _ptr(w), _ptr(M), _ptr(B), _ptr(C_), mask)
        self._check(rc, "raftx_dense_resident")
        self._dense = (nS, n, nw)

    def solve_dense_resident(self, F, Badd=None, want_Z=False):
        """Xi [nSys,nRhs,n,nw] (and Z [nSys,n,n,nw]) with the resident matrices: nSys = nSet * nPer systems, system s with
        the matrices of unit s // nPer and B + Badd[s] (Badd [nSys,n,n] or None): raftx_solve_dense_resident."""
        if getattr(self, "_dense", None) is None:
            raise ValueError("solve_dense_resident: dense_resident first")
        nSet, n, nw = self._dense
        F = _c128(F)
        nS, nR = F.shape[0], F.shape[1]
        if F.shape != (nS, nR, n, nw) or nS % nSet:
            raise ValueError("F must be [nSet*nPer,nRhs,n,nw]")
        Badd = None if Badd is None else _f64(Badd, (nS, n, n), "Badd")
        Xi = np.empty((nS, nR, n, nw), dtype=np.complex128)
        Z = np.empty((nS, n, n, nw), dtype=np.complex128) if want_Z else None
        rc = self.rlib.lib.raftx_solve_dense_resident(self._h, nS // nSet, _ptr(Badd), nR, _ptr(F), _ptr(Xi), _ptr(Z))
        self._check(rc, "raftx_solve_dense_resident")
        return (Xi, Z) if want_Z else Xi

    def flex_solve(self, node_off, Tn, M, B, C_, F_lin, nIter, tol, XiStart, want_B=True, want_F=True, want_Z=False, out=None):
        """The fixed point of units with more than 6 reduced DOFs on the resident node tables and sea states
        (raftx_flex_solve): node_off [nUnit+1], Tn [nNode,6,n], M, B [nUnit,n,n(,nw)], C_ [nUnit,n,n], F_lin
        [nUnit,nCase,nHead,n,nw].  dict Xi [nUnit,nCase,nHead,n,nw], niter, flags [nUnit,nCase], B_drag, F_drag, Z.
        out: optional dict of preallocated arrays for "Xi", "B_drag", "F_drag", "Z" (e.g. from pinned_empty)."""
        node_off = np.ascontiguousarray(node_off, dtype=np.int64)
        nU = len(node_off) - 1
        Tn = _f64(Tn)
        n = Tn.shape[2]
        F_lin = _c128(F_lin)
        nC, nH, nw = F_lin.shape[1], F_lin.shape[2], F_lin.shape[4]
        if Tn.shape != (int(node_off[-1]), 6, n) or F_lin.shape != (nU, nC, nH, n, nw):
            raise ValueError("flex_solve: Tn must be [nNode,6,n] and F_lin [nUnit,nCase,nHead,n,nw]")
        M, B = _f64(M), _f64(B)
        mask = 0
        for bit, A, name in ((1, M, "M"), (2, B, "B")):
            if A.shape == (nU, n, n, nw):
                mask |= bit
            elif A.shape != (nU, n, n):
                raise ValueError("%s must be [nUnit,n,n] or [nUnit,n,n,nw]" % name)
        C_ = _f64(C_, (nU, n, n), "C")
        out = out or {}

        def buf(key, shape, dtype, want):                 # a caller's preallocated (e.g. page-locked) array, or a fresh one
            if not want:
                return None
            a = out.get(key)
            if a is None:
                return np.empty(shape, dtype=dtype)
            if a.shape != tuple(shape) or a.dtype != np.dtype(dtype) or not a.flags["C_CONTIGUOUS"]:
                raise ValueError("flex_solve: out[%r] must be a C-contiguous %s array of shape %s" % (key, np.dtype(dtype), tuple(shape)))
            return a
        Xi = buf("Xi", (nU, nC, nH, n, nw), np.complex128, True)
        niter = np.zeros((nU, nC), dtype=np.int32)
        flags = np.zeros((nU, nC), dtype=np.int32)
        Bd = buf("B_drag", (nU, nC, n, n), np.float64, want_B)
        Fd = buf("F_drag", (nU, nC, nH, n, nw), np.complex128, want_F)
        Z = buf("Z", (nU, nC, n, n, nw), np.complex128, want_Z)
        rc = self.rlib.lib.raftx_flex_solve(self._h, nU, _ptr(node_off), n, _ptr(Tn), _ptr(M), _ptr(B), _ptr(C_), mask, _ptr(F_lin),
                                            int(nIter), float(tol), float(XiStart), _ptr(Xi), _ptr(niter), _ptr(flags), _ptr(Bd),
                                            _ptr(Fd), _ptr(Z))
        self._check(rc, "raftx_flex_solve")
        return {"Xi": Xi, "niter": niter, "flags": flags, "B_drag": Bd, "F_drag": Fd, "Z": Z}

    def flex_start(self, XiLast0):
        """The iterate the NEXT flex_solve starts from instead of XiStart (raftx_flex_start; one-shot): XiLast0
        [nUnit,nCase,n,nw], or None to clear."""
        if XiLast0 is None:
            self._check(self.rlib.lib.raftx_flex_start(self._h, 0, 0, None), "raftx_flex_start")
            return
        X = _c128(XiLast0)
        if X.ndim != 4:
            raise ValueError("flex_start: XiLast0 must be [nUnit,nCase,n,nw]")
        self._check(self.rlib.lib.raftx_flex_start(self._h, X.shape[0], X.shape[2], _ptr(X)), "raftx_flex_start")

    def debug_flex_gemm(self, A, W):
        """A^T W ([K,n] each, K a multiple of 6) through the projection kernel of raftx_flex_solve (test hook)."""
        A, W = _f64(A), _f64(W)
        K, n = A.shape
        out = np.empty((n, n))
        self._check(self.rlib.lib.raftx_debug_flex_gemm(self._h, K, n, _ptr(A), _ptr(W), _ptr(out)), "raftx_debug_flex_gemm")
        return out

    def synchronize(self):
        """hipDeviceSynchronize on the context's device"""
        self._check(self.rlib.lib.raftx_device_synchronize(self._h), "raftx_device_synchronize")

    def last_solve_kernel(self):
        """(feature bits, waves per SIMD, run-start cache slots) of the fused kernel the last solve launched."""
        f, w, n = C.c_int(0), C.c_int(0), C.c_int(0)
        rc = self.rlib.lib.raftx_last_solve_kernel(self._h, C.byref(f), C.byref(w), C.byref(n))
        self._check(rc, "raftx_last_solve_kernel")
        return f.value, w.value, n.value

    def debug_math(self, x, table=False):
        """The device's own sincos / exp on x; table=True: the table-driven sincos of the fused kernel's run starts."""
        x = _f64(x).ravel()
        s, c, e = np.empty_like(x), np.empty_like(x), np.empty_like(x)
        fn = self.rlib.lib.raftx_debug_math_table if table else self.rlib.lib.raftx_debug_math
        rc = fn(self._h, len(x), _ptr(x), _ptr(s), _ptr(c), _ptr(e))
        self._check(rc, "raftx_debug_math")
        return s, c, e

    # ------------------------------------------------------------- multi-GPU exchange steps (RCCL, raftx_comm_*)
    def comm_unique_id(self):
        buf = C.create_string_buffer(128)
        self._check(self.rlib.lib.raftx_comm_unique_id(self._h, buf), "raftx_comm_unique_id")
        return buf.raw

    def comm_init(self, rank, world, unique_id):
        if len(unique_id) != 128:
            raise ValueError("the RCCL unique id is 128 bytes")
        self._check(self.rlib.lib.raftx_comm_init(self._h, int(rank), int(world), C.c_char_p(bytes(unique_id))), "raftx_comm_init")
        self._comm = (int(rank), int(world))

    def _need_comm(self, what):
        if self._comm is None:
            raise RaftxError("%s: no communicator on this context (comm_init first)" % what)
        return self._comm

    def comm_destroy(self):
        self._check(self.rlib.lib.raftx_comm_destroy(self._h), "raftx_comm_destroy")
        self._comm = None

    def comm_broadcast(self, arr, root=0):
        """In place: ``arr`` (C-contiguous NumPy array) is sent by ``root`` and overwritten on every other rank."""
        if not arr.flags["C_CONTIGUOUS"]:
            raise ValueError("comm_broadcast needs a C-contiguous array")
        self._check(self.rlib.lib.raftx_comm_broadcast(self._h, _ptr(arr), arr.nbytes, int(root)), "raftx_comm_broadcast")
        return arr

    def comm_gather_rows(self, local, counts, root=0):
        """Row blocks of every rank (counts[r] rows each, same trailing shape and dtype) back to back on root."""
        local = np.ascontiguousarray(local)
        counts = np.ascontiguousarray(counts, dtype=np.int64)
        rank, _ = self._need_comm("comm_gather_rows")
        if local.shape[0] != counts[rank]:
            raise ValueError("this rank holds %d rows, counts says %d" % (local.shape[0], counts[rank]))
        row_bytes = int(np.prod(local.shape[1:], dtype=np.int64)) * local.dtype.itemsize
        out = np.empty((int(counts.sum()),) + local.shape[1:], dtype=local.dtype) if rank == root else None
        self._check(self.rlib.lib.raftx_comm_gather_rows(self._h, _ptr(local), _ptr(counts), row_bytes, _ptr(out), int(root)),
                    "raftx_comm_gather_rows")
        return out

    def comm_gather_xi(self, counts, root=0, out=None):
        """The resident responses of every rank's last solve -> [sum(counts), nHead, 6, nw] on root, HBM to HBM."""
        counts = np.ascontiguousarray(counts, dtype=np.int64)
        rank, _ = self._need_comm("comm_gather_xi")
        if rank == root:
            shape = (int(counts.sum()), self.nHead, 6, self.nw)
            if out is None:
                out = np.empty(shape, dtype=np.complex128)
            elif out.dtype != np.complex128 or out.size != int(np.prod(shape)) or not out.flags["C_CONTIGUOUS"]:
                raise ValueError("out must be a C-contiguous complex128 array with %d elements" % int(np.prod(shape)))
        self._check(self.rlib.lib.raftx_comm_gather_xi(self._h, _ptr(counts), _ptr(out) if rank == root else None, int(root)),
                    "raftx_comm_gather_xi")
        return out if rank == root else None

    def comm_reduce_sum(self, buf, root=0):
        """In place on root: element-wise sum over ranks of a float64 array."""
        if buf.dtype != np.float64 or not buf.flags["C_CONTIGUOUS"]:
            raise ValueError("comm_reduce_sum needs a C-contiguous float64 array")
        self._check(self.rlib.lib.raftx_comm_reduce_sum(self._h, _ptr(buf), buf.size, int(root)), "raftx_comm_reduce_sum")
        return buf

    def last_kernel_ms(self):
        return float(self.rlib.lib.raftx_last_kernel_ms(self._h))
