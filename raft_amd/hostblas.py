"""The host side of a call is a handful of small dense products (reductions with a unit's T, zero tests of rotor tables).
On a many-core host OpenBLAS starts all its threads for each of them and they keep spinning afterwards, beside the runtime
threads of the device library -- and where the process runs under a CPU quota (the pool's GPU boxes: 256 logical CPUs, a cgroup
quota of 16) the spinning threads spend the quota and EVERYTHING in the process is throttled for the rest of the scheduler
period: measured there, a drop-in call of the flexible deck takes 24 ms instead of 6 (a 150 x 150 addition 2.8 ms), FlexSweep.run
40-70 ms every few calls instead of 24 (scripts/prof_dropin_lines.py, scripts/prof_flex_batch.py).  ``few_threads()`` scopes
the products of this package to at most eight BLAS threads.  It only ever LOWERS the count: a user who runs with
OPENBLAS_NUM_THREADS=1 keeps one thread.  Without threadpoolctl it does nothing."""
import contextlib
import os

try:
    from threadpoolctl import ThreadpoolController
except ImportError:                                  # optional
    ThreadpoolController = None
_BLAS = None                                         # the controller, made once (it walks the loaded libraries)
_LIMIT = None


def _limit():
    global _LIMIT
    if _LIMIT is None:
        try:
            n = len(os.sched_getaffinity(0))
        except (AttributeError, OSError):
            n = os.cpu_count() or 1
        _LIMIT = max(1, min(8, n))
    return _LIMIT


def few_threads():
    """Context manager: BLAS products inside run on at most min(8, CPUs of this process) threads (3-6 us to enter and leave)."""
    global _BLAS
    if ThreadpoolController is None:
        return contextlib.nullcontext()
    if _BLAS is None:
        _BLAS = ThreadpoolController()
    lim = _limit()
    now = [m.get("num_threads") for m in _BLAS.info() if m.get("user_api") == "blas" and m.get("num_threads")]
    if now and max(now) <= lim:
        return contextlib.nullcontext()
    return _BLAS.limit(limits=lim, user_api="blas")
