"""Host-side helpers of the drop-in (raft_amd/dropin.py, raft_amd/hostblas.py): the real-times-complex product on the interleaved
view, the fast zero test of large tables, the BLAS thread scope."""
import numpy as np
import pytest

from raft_amd import dropin, hostblas


def test_real_times_complex_equals_the_mixed_product():
    rng = np.random.default_rng(0)
    for shape_T, shape_X in (((216, 6), (2, 6, 200)), ((150, 360), (3, 360, 40)), ((6, 6), (6, 7)), ((5, 3), (4, 2, 3, 9))):
        T = rng.normal(size=shape_T)
        X = rng.normal(size=shape_X) + 1j * rng.normal(size=shape_X)
        Y = dropin._real_times_complex(T, X)
        ref = np.matmul(T.astype(complex), X)
        assert Y.shape == ref.shape and Y.dtype == np.complex128 and Y.flags.c_contiguous
        assert np.abs(Y - ref).max() <= 1e-13 * np.abs(ref).max()
    big = rng.normal(size=(3, 12, 50)) + 1j * rng.normal(size=(3, 12, 50))
    view = big[:, 3:9, :]                                   # a non-contiguous slice (model.Xi[:, i*6:(i+1)*6, :] of a farm)
    T = rng.normal(size=(20, 6))
    assert np.allclose(dropin._real_times_complex(T, view), np.matmul(T.astype(complex), view), rtol=1e-13, atol=0)
    assert np.array_equal(big[:, 3:9, :], view)             # the operand is not written to


def test_nonzero_is_np_any_for_large_real_tables():
    z = np.zeros((150, 150, 40))
    assert dropin._nonzero(z) is False
    a = z.copy(); a[17, 3, 5] = 1e-3
    assert dropin._nonzero(a) is True
    tiny = z.copy(); tiny[0, 0, 0] = 1e-200                 # its square underflows: the bit patterns decide
    assert dropin._nonzero(tiny) is True
    neg0 = z.copy(); neg0[1, 1, 1] = -0.0                   # counts as something (adding it changes nothing)
    assert dropin._nonzero(neg0) is True
    nan = z.copy(); nan[2, 2, 2] = np.nan
    assert dropin._nonzero(nan) is True
    for small in (np.zeros((6, 6, 200)), np.eye(6), np.zeros(3, dtype=complex), 0.0, np.array(2.0)):
        assert dropin._nonzero(small) == bool(np.any(small))
    assert dropin._nonzero(z[:, :, ::2]) is False           # not contiguous: np.any itself


def test_blas_scope_only_lowers_the_thread_count():
    pytest.importorskip("threadpoolctl")
    from threadpoolctl import ThreadpoolController
    ctl = ThreadpoolController()
    blas = [m for m in ctl.info() if m.get("user_api") == "blas"]
    if not blas:
        pytest.skip("no BLAS library seen by threadpoolctl")
    before = max(m["num_threads"] for m in blas)
    with hostblas.few_threads():
        inside = max(m["num_threads"] for m in ThreadpoolController().info() if m.get("user_api") == "blas")
        assert inside <= max(1, min(8, before))
    assert max(m["num_threads"] for m in ThreadpoolController().info() if m.get("user_api") == "blas") == before
    with ctl.limit(limits=1, user_api="blas"):              # a user who runs single-threaded keeps one thread
        with hostblas.few_threads():
            assert max(m["num_threads"] for m in ThreadpoolController().info() if m.get("user_api") == "blas") == 1
