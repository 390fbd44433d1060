"""Host-side helpers shared by the pipeline and the drop-in mirrors."""
_blas_controller = None


class _NoLimit:
    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return False


def single_threaded_blas(n=0):
    """Context for the host tail on the n x n triangle (n <= 512): BLAS / LAPACK on one thread (8 us to enter and leave).
    A 185 x 185 triangular inverse is 2 Mflop, but a threaded BLAS wakes every core it sees for it and its idle threads
    spin; on a host whose CPU time is capped (container quota: 16 CPUs of 256 visible on the GPU boxes used here) that
    exhausts the quota and the whole process -- the thread waiting for the GPU included -- is frozen until the next
    100 ms accounting period.  Measured (tools/step_trace.py, tools/trace_gaps.sh, /sys/fs/cgroup/cpu.stat): TIAGo steps of
    exactly 100.0 ms for 71 ms of kernels, TALOS steps wandering between 218 and 280 ms, `nr_throttled` 0 -> 14 in one
    bench run; with the limit the steps equal the kernels (71 ms, 218 ms)."""
    global _blas_controller
    if _blas_controller is None:
        try:
            import scipy.linalg.lapack  # noqa: F401  (SciPy carries its own BLAS: it must be loaded before the controller looks)
            from threadpoolctl import ThreadpoolController
            _blas_controller = ThreadpoolController()
        except Exception:  # noqa: BLE001  (threadpoolctl is optional)
            _blas_controller = False
    if not _blas_controller:
        return _NoLimit()
    return _blas_controller.limit(limits=1, user_api="blas")


def host_tail(fn):
    """Decorator for the drop-in functions whose host part is small dense algebra on n x n triangles (inverse, SVD, pivoted
    QR, QP): that part runs with BLAS / LAPACK on one thread (see single_threaded_blas); the device work is unaffected."""
    import functools

    @functools.wraps(fn)
    def wrapper(*args, **kwargs):
        with single_threaded_blas():
            return fn(*args, **kwargs)

    return wrapper
