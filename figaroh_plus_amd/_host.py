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


def cpu_budget():
    """CPUs this process may actually burn: the scheduler affinity capped by the cgroup quota (cpu.max) when there is one
    -- the GPU boxes show 256 logical CPUs under a quota of 16, and a process that exceeds it is frozen (see above)."""
    import os
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            quota, period = f.read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(float(quota) / float(period))))
    except (OSError, ValueError):
        pass
    return max(1, n)


def singular_values_batch(R):
    """Singular values of a stack of n x n matrices (B x n x n) -> B x n, descending.  The B factorisations are
    independent LAPACK calls that release the interpreter lock: with BLAS on one thread each they run on a small pool of
    threads, half the CPU budget at most (a 180 x 180 SVD is about 1 ms, 64 of them would otherwise be the whole cost of
    a batched objective whose device part takes 7.5 ms)."""
    import numpy as np
    R = np.asarray(R)
    B = R.shape[0]
    workers = min(B, max(1, cpu_budget() // 2), 8)
    if workers <= 1 or B < 4:
        return np.linalg.svd(R, compute_uv=False)
    from concurrent.futures import ThreadPoolExecutor
    out = np.empty(R.shape[:2])

    def work(lo, hi):
        with single_threaded_blas():
            out[lo:hi] = np.linalg.svd(R[lo:hi], compute_uv=False)

    bounds = np.linspace(0, B, workers + 1).astype(int)
    with ThreadPoolExecutor(workers) as pool:
        list(pool.map(lambda k: work(bounds[k], bounds[k + 1]), range(workers)))
    return out


def relative_percent(sigma, phi):
    """100 sigma_i / |phi_i| rounded to two decimals -- the reference's std% (identification_tools.py:226-232,
    examples/staubli_TX40/identification.py:342-346).  An estimate that rounds to exactly zero (the scripts round phi to six
    decimals first) gives ``inf`` there as well -- NumPy's division by zero, with a RuntimeWarning per element; here the same
    ``inf`` is returned on purpose and without the warning: "this parameter's relative uncertainty is unbounded" is what the
    essential-parameter loop (:354-399) acts on when it drops the parameter with the largest std%."""
    import numpy as np
    sigma = np.asarray(sigma, dtype=np.float64)
    phi = np.abs(np.asarray(phi, dtype=np.float64))
    out = np.full(sigma.shape, np.inf)
    np.divide(100.0 * sigma, phi, out=out, where=phi != 0.0)
    out[(phi == 0.0) & (sigma == 0.0)] = np.nan  # (0 / 0, as NumPy has it)
    return np.round(out, 2)
