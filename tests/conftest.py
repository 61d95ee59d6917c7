import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

ORACLE_SO = os.path.join(ROOT, "oracle", "libraftx_oracle.so")
HIP_SO = os.path.join(ROOT, "raft_amd", "csrc", "libraftx_hip.so")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # "-m gpu" asks for the device tests: without a GPU they must FAIL.  Any other selection on a box without a GPU
    # (a plain "pytest tests/") skips them instead of erroring at fixture set-up.
    expr = (config.getoption("-m") or "").strip()
    config._raftx_require_gpu = (expr == "gpu") or os.environ.get("RAFTX_REQUIRE_GPU") == "1"


def _build_oracle():
    srcs = [os.path.join(ROOT, "oracle", f) for f in ("raftx_oracle.c", "raftx_geom_oracle.h")] + \
           [os.path.join(ROOT, "include", "raftx.h")]
    if (not os.path.exists(ORACLE_SO)) or any(os.path.getmtime(ORACLE_SO) < os.path.getmtime(s) for s in srcs):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")], stdout=subprocess.DEVNULL)
    return ORACLE_SO


@pytest.fixture(scope="session")
def oracle_lib():
    """The CPU oracle bound through the same ctypes class as the product."""
    from raft_amd._abi import RaftxLib
    return RaftxLib(_build_oracle())


@pytest.fixture()
def oracle_ctx(oracle_lib):
    ctx = oracle_lib.context(0)
    yield ctx
    ctx.close()


def _gpu_present():
    """Does raftx_ctx_create find a device?  (cached; the library itself loads anywhere)"""
    if not hasattr(_gpu_present, "v"):
        from raft_amd import backend
        from raft_amd._abi import RaftxError
        try:
            backend.hip_library().context(0).close()
            _gpu_present.v = True
        except RaftxError as e:
            if "rc=-3" not in str(e):
                raise
            _gpu_present.v = False
    return _gpu_present.v


@pytest.fixture(autouse=True)
def _skip_device_tests_without_a_gpu(request):
    """Every @pytest.mark.gpu test -- whatever fixtures it uses -- is skipped on a box without a GPU unless the device
    tests were asked for (-m gpu / RAFTX_REQUIRE_GPU=1), in which case they run and FAIL."""
    if request.node.get_closest_marker("gpu") is not None and not request.config._raftx_require_gpu and not _gpu_present():
        pytest.skip("no GPU on this box (run the device tests with -m gpu on an MI355X)")


@pytest.fixture(scope="session")
def hip_lib():
    """The product library; GPU tests fail (not skip) if it is missing."""
    from raft_amd import backend
    return backend.hip_library()


@pytest.fixture()
def hip_ctx(hip_lib, request):
    from raft_amd._abi import RaftxError
    try:
        ctx = hip_lib.context(0)
    except RaftxError as e:
        if "rc=-3" in str(e) and not request.config._raftx_require_gpu:          # raftx_ctx_create: no GPU on this box
            pytest.skip("no GPU on this box (run the device tests with -m gpu on an MI355X)")
        raise
    yield ctx
    ctx.close()
