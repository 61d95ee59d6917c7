"""Loader of the product library: raft_amd/csrc/libraftx_hip.so (gfx950).

There is NO CPU fallback.  If the HIP extension is missing, or no MI355X is
visible, every entry point raises -- loudly -- instead of computing on the
host.  (The CPU oracle under oracle/ is test infrastructure; only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg may bind it, and they
do so explicitly through raft_amd._abi.RaftxLib, never through this module.)
"""
import os
import threading

from ._abi import RaftxLib, RaftxError

HIP_LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc", "libraftx_hip.so")

_lock = threading.Lock()
_lib = None
_ctx = {}


def hip_library():
    """The bound product library (built by __graft_entry__.build())."""
    global _lib
    with _lock:
        if _lib is None:
            path = os.environ.get("RAFTX_HIP_LIB", HIP_LIB_PATH)      # tuning builds only; must still be a device library
            if path != HIP_LIB_PATH:
                lib = RaftxLib(path)
                if not lib.is_device:
                    raise RaftxError("%s is not a device library" % path)
                _lib = lib
                return _lib
            if not os.path.exists(HIP_LIB_PATH):
                raise RaftxError(
                    "HIP extension not built: %s is missing. Run "
                    "`python -c 'import __graft_entry__ as g; g.build()'` (hipcc --offload-arch=gfx950). "
                    "raft_amd has no CPU fallback." % HIP_LIB_PATH)
            lib = RaftxLib(HIP_LIB_PATH)
            if not lib.is_device:
                raise RaftxError("%s is not a device library" % HIP_LIB_PATH)
            _lib = lib
        return _lib


def default_context(device_id=None):
    """Process-wide ctx for ``device_id`` (default: LOCAL_RANK or 0)."""
    if device_id is None:
        device_id = int(os.environ.get("LOCAL_RANK", "0"))
    lib = hip_library()
    with _lock:
        if device_id not in _ctx:
            _ctx[device_id] = lib.context(device_id)
        return _ctx[device_id]
