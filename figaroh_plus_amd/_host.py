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


NULL_RULE_FOLD = 64.0  # the null-pivot rule folds at most tol_qr / 64 of a column's norm (include/figh.h)


def null_rule_bounds(R1, R2):
    """(A_base, A_dep) for :func:`null_rule_certified` from the regrouped factorisation ``qr([W1 W2]) = [R1 R2; 0 ~0]``:
    how much a perturbation of the earlier columns can move a column's pivot.  ``A_dep[c] = sum_i |beta_ic|`` (the dependent
    column is ``sum_i beta_ic b_i``), ``A_base[k] = |R1_kk| sum_{i<k} |(R1^-1)_ik|`` (= ``sum |R1[:k,:k]^-1 R1[:k,k]|``: the
    coefficients of base column k's projection on the base columns in front of it).  None when R1 is singular."""
    import numpy as np
    from scipy.linalg.lapack import dtrtri
    r = R1.shape[0]
    if r == 0:
        return np.zeros(0), np.zeros(R2.shape[1] if R2.ndim == 2 else 0), 0.0
    if not np.diag(R1).all():
        return None
    X, info = dtrtri(np.ascontiguousarray(R1), lower=0)
    if info != 0 or not np.isfinite(X).all():
        return None
    X = np.triu(X)
    A_base = np.abs(np.diag(R1)) * (np.abs(X).sum(axis=0) - np.abs(np.diag(X)))
    A_dep = np.abs(X @ R2).sum(axis=0) if R2.size else np.zeros(0)
    return A_base, A_dep, float(np.abs(X).sum(axis=1).max())  # (the last: the infinity norm of R1^-1, for the bound on phi)


def null_rule_certified(absdiag, idx_base, idx_regroup, bounds, tol_qr, safety=2.0, phi=None):
    """A-posteriori GUARD for a factorisation that ran WITH the null-pivot rule (``figh_tsqr_null_pivot_tol``): True when the
    classification ``|R_kk| > tol_qr`` can be trusted to be the one plain Householder -- the reference's ``np.linalg.qr``,
    qrdecomposition.py:205-221 -- gives on the same matrix; not certified = the caller repeats the factorisation WITHOUT the
    rule, so that what it returns is the reference's classification in every case.

    The rule makes R the exact factor of ``W + E``: in every level-0 triangle at most ``tol_qr / 64`` of a column's norm is
    folded without a reflector, so the later columns keep their component along that direction.  A pivot is the distance of
    its column from the span of the columns in front of it; to first order ``|R_kk|`` moves by at most ``(1 + A_k) |E|``, A_k =
    the 1-norm of the coefficients that express the column (dependent) or its projection (base) through the base columns in
    front of it (:func:`null_rule_bounds`).  Three conditions, all on the numbers of the pass itself:

    * every base pivot lies further above ``tol_qr`` and every dependent pivot further below it than
      ``safety (1 + A_k) tol_qr / 64``: no decision sits within reach of the perturbation, and LARGE REGROUPING COEFFICIENTS --
      the failure tools/fuzz_trees.py found in round 6: random trees with cond(W_b) ~ 1e14, coefficients in the hundreds, a
      spurious base pivot of 1.1e-7 -- fail it (a spurious base column shows as a tiny ``R1_kk`` with a huge ``A_k``);
    * no dependent pivot above ``tol_qr / 8``: a column that is classified dependent but carries that much genuine content of
      its own (TIAGo at 4e5 samples: 6.7e-9) may have had ALL of it folded -- over T level-0 triangles the folded parts add up
      to ``sqrt(T) tol_qr / 64`` in the worst case, include/figh.h -- and its direction then leaks into the columns behind it;
      the rounding residue of an exactly dependent column stays far below (1e-12 .. 1.5e-10 at 1e7 samples);
    * with ``phi`` (the least-squares solution of the pass): the solution itself is within reach of the perturbation when
      ``|R1^-1|_inf (tol_qr / 64) |phi|_1 > 1e-7 max(1, |phi|_inf)`` -- phi solves the problem of ``W_b + E_b`` instead of ``W_b``,
      ``|d phi|_inf <= |R1^-1|_inf |E_b phi|`` -- a tenth of north_star's 1e-6 on the estimates.  Plain Householder's own
      error is ``cond eps``; the rule's is ``cond tol_qr / (64 |W_b|)``, some 1e4 times that, harmless for the BASELINE robots
      (cond(W_b) 1e2 .. 1e4) and not for an ill-conditioned base regressor (found by tools/fuzz_trees.py: two layouts of one
      random model, one under the rule and one not, residuals 1e-5 apart);
    * finite numbers throughout.

    For the five BASELINE robots A <= 9 and UR10, TALOS and the human model are certified at their full sizes; TIAGo, whose
    regressor has genuine pivots within 6 % of ``tol_qr`` at 1e6 samples, is not and runs without the rule.  A guard, not a
    proof: the bound uses the per-triangle folding limit for |E| (a column that folds its maximum in EVERY triangle has a
    residual of its own near the tolerance scale and trips the second condition)."""
    import numpy as np
    if bounds is None:
        return False
    A_base, A_dep, xnorm = bounds
    fold = safety * tol_qr / NULL_RULE_FOLD
    if phi is not None:
        phi = np.abs(np.asarray(phi, dtype=np.float64))
        if not np.isfinite(phi).all() or xnorm * (tol_qr / NULL_RULE_FOLD) * phi.sum() > 1e-7 * max(1.0, phi.max(initial=0.0)):
            return False
    d = np.abs(np.asarray(absdiag, dtype=np.float64))
    db, dd = d[np.asarray(idx_base, dtype=np.int64)], d[np.asarray(idx_regroup, dtype=np.int64)]
    if not (np.isfinite(db).all() and np.isfinite(dd).all()):
        return False
    if dd.size and dd.max() > tol_qr / 8.0:
        return False
    return bool(np.all(db - tol_qr > (1.0 + A_base) * fold) and np.all(tol_qr - dd > (1.0 + A_dep) * fold))
