"""HIP-backed mirror of ``src/figaroh/tools/qrdecomposition.py:24-332``.

The reference factorises the tall matrix twice with ``np.linalg.qr`` (forming Q)
and then only looks at R.  Here the rows of ``W_e`` are streamed once through the
Householder TSQR kernel (``figh_tsqr``), which returns the n x n triangle (and
``Q^T tau`` as an extra column).  The regrouped factorisation
``qr([W1 W2])`` equals ``qr(R[:, perm])`` up to row signs, so the second QR runs
on the permuted triangle (same kernel, one wavefront); ``beta = R1^-1 R2`` and
``phi_b = R1^-1 Q1^T tau`` are invariant to those signs.  Selection, rounding
and the expression strings follow the reference statement by statement.
"""
import numpy as np

from .. import _lib
from .._host import host_tail
from .._host import null_rule_bounds, null_rule_certified
from .._host import single_threaded_blas
from ..device import GpuMatrix, index_to_device, to_device, vector_to_device

TOL_QR = 1e-8


def rfactor(W, tau=None, col_idx=None, block_weight=None, tol_qr=None):
    """Upper-triangular factor of ``[W[:, col_idx], tau]`` (rows scaled per block by ``block_weight``).

    Returns an (nc, nc) array, nc = n (+1 with tau); with tau the last column holds Q^T tau and,
    in its last entry, the residual norm of the least-squares problem.  ``tol_qr``: the caller is going to
    classify |R_kk| <= tol_qr as dependent -- columns that are null to tol_qr / 64 skip their column steps
    (``_lib.null_pivots``); None: plain Householder.
    """
    Wd, _ = to_device(W)
    if Wd.rows == 0:
        raise ValueError("empty regressor")
    n = Wd.cols if col_idx is None else len(col_idx)
    d_idx = None if col_idx is None else index_to_device(col_idx)
    d_tau = None
    if tau is not None:
        if hasattr(tau, "ptr"):  # a device vector (e.g. reject_rows on a device-resident W): stays where it is
            ntau = int(tau.size)
        else:
            tau = np.ascontiguousarray(tau, dtype=np.float64).reshape(-1)
            ntau = tau.shape[0]
        if ntau != Wd.rows:
            raise ValueError("tau has %d entries, W has %d rows" % (ntau, Wd.rows))
        d_tau = vector_to_device(tau)
    nc = n + (1 if tau is not None else 0)
    d_R = _lib.DeviceArray((nc * nc,), np.float64)
    with _lib.null_pivots(tol_qr):
        _lib.tsqr(Wd.buf, Wd.rows, Wd.ld, d_idx, n, d_tau, block_weight, d_R)
    return np.triu(d_R.to_host().reshape(nc, nc))


def _select(diagR, params_r, tol_qr):
    assert diagR.shape[0] == len(params_r), "params_r does not have same length with R"
    big = np.abs(np.asarray(diagR)) > tol_qr  # qrdecomposition.py:215-221, vectorised (NaN: regrouped, as in the loop)
    return np.flatnonzero(big).tolist(), np.flatnonzero(~big).tolist()


def _regroup(R, idx_base, idx_regroup, with_tau):
    """qr([W1 W2 (tau)]) from the triangle of qr([W_e (tau)]): returns (R1, R2, Q1^T tau | None)."""
    n = len(idx_base) + len(idx_regroup)
    perm = list(idx_base) + list(idx_regroup) + ([n] if with_tau else [])
    R_r = rfactor(np.ascontiguousarray(R[:, perm]))
    r = len(idx_base)
    return R_r[:r, :r], R_r[:r, r:n], (R_r[:r, n] if with_tau else None)


def _expressions(params_base, params_regroup, beta, tol_beta=1e-6):
    """qrdecomposition.py:246-266: ``base + " + " / " - " + str(abs(beta)) + "*" + regrouped`` for abs(beta) >= tol,
    terms in column order.  Only the non-negligible entries are visited (np.nonzero is row-major, so the order is the
    reference's double loop); str() of a Python float and of a NumPy float64 are the same shortest repr."""
    out = list(params_base)
    beta = np.asarray(beta)
    ii, jj = np.nonzero(~(np.abs(beta) < tol_beta))
    vals = beta[ii, jj]
    neg = (vals < -tol_beta).tolist()
    mags = np.abs(vals).tolist()
    names = [str(p) for p in params_regroup]
    for i, j, m, ng in zip(ii.tolist(), jj.tolist(), mags, neg):
        out[i] += (" - " if ng else " + ") + str(m) + "*" + names[j]
    return out


def _base_columns(Wd, idx, keep_on_device=False):
    out = GpuMatrix.empty(Wd.rows, len(idx))
    if len(idx):
        _lib.gather_cols(Wd.buf, Wd.rows, Wd.ld, index_to_device(idx), len(idx), out.buf, len(idx))
    return out if keep_on_device else out.numpy()


def _factor_and_select(Wd, params_r, tol_qr, null_pivots, tau=None):
    """(R, idx_base, idx_regroup, R1, R2, Q1^T tau | None): the triangle of ``[W_e (tau)]``, the selection
    (qrdecomposition.py:215-221) and the regrouped factorisation.  With ``null_pivots`` the factorisation runs under the
    null-pivot rule and its classification is CERTIFIED against plain Householder afterwards (``_host.null_rule_certified``);
    a matrix for which the certificate does not hold -- pivots close to ``tol_qr``, or regrouping coefficients so large that
    the folded ``tol_qr / 64`` could matter -- is factored again without the rule: what comes back is always the
    classification of the reference's arithmetic."""
    n = len(params_r)
    for rule in ((True, False) if null_pivots else (False,)):
        R = rfactor(Wd, tau=tau, tol_qr=tol_qr if rule else None)
        assert R.shape[0] == n + (1 if tau is not None else 0), "params_r does not have same length with R"
        idx_base, idx_regroup = _select(np.diag(R)[:n], params_r, tol_qr)
        R1, R2, q1t_tau = _regroup(R, idx_base, idx_regroup, tau is not None)
        if not rule:
            break
        with single_threaded_blas():
            bounds = null_rule_bounds(R1, R2)
        phi = None
        if q1t_tau is not None and bounds is not None and len(idx_base):
            with single_threaded_blas():
                phi = np.linalg.solve(np.triu(R1), q1t_tau)  # (np.linalg.inv(R1) @ q1t_tau of the callers, for the bound on phi)
        if null_rule_certified(np.diag(R)[:n], idx_base, idx_regroup, bounds, tol_qr, phi=phi):
            break
    return R, idx_base, idx_regroup, R1, R2, q1t_tau


@host_tail
def get_baseIndex(W_e, params_r, tol_qr=TOL_QR, null_pivots=True):
    """Indices of the linearly independent columns (qrdecomposition.py:274-296).  ``null_pivots=False`` (not a reference
    argument): plain Householder steps on every column, the reference's LAPACK arithmetic step for step, instead of the
    null-pivot rule of include/figh.h (columns that are zero to tol_qr / 64 below the triangle skip their reflector).  With
    the rule the classification is certified afterwards and the factorisation repeated without it when the certificate does
    not hold (:func:`_factor_and_select`): the index set is the reference's either way."""
    Wd, _ = to_device(W_e)
    _, idx_base, _, _, _, _ = _factor_and_select(Wd, params_r, tol_qr, null_pivots)
    return tuple(idx_base)


def build_baseRegressor(W_e, idx_base):
    """Columns ``idx_base`` of ``W_e`` (qrdecomposition.py:299-313)."""
    Wd, on_dev = to_device(W_e)
    return _base_columns(Wd, list(idx_base), on_dev)


@host_tail
def get_baseParams(W_e, params_r, params_std=None, tol_qr=TOL_QR, null_pivots=True):
    """(W_b, params_base, idx_base) -- qrdecomposition.py:190-271.  ``null_pivots``: see :func:`get_baseIndex`."""
    Wd, on_dev = to_device(W_e)
    _, idx_base, idx_regroup, R1, R2, _ = _factor_and_select(Wd, params_r, tol_qr, null_pivots)
    with single_threaded_blas():  # n x n host work: see _host.py
        beta = np.around(np.matmul(np.linalg.inv(R1), R2), 6)
    params_base = _expressions([params_r[i] for i in idx_base], [params_r[i] for i in idx_regroup], beta)
    # W_b = Q1 R1 is by construction the gathered base columns (the reference asserts it, :268-269)
    W_b = _base_columns(Wd, idx_base, on_dev)
    return W_b, params_base, idx_base


@host_tail
def double_QR(tau, W_e, params_r, params_std=None, tol_qr=TOL_QR, null_pivots=True):
    """(W_b, base_parameters, params_base, phi_b[, phi_std]) -- qrdecomposition.py:89-187.  ``null_pivots``: see
    :func:`get_baseIndex`."""
    Wd, on_dev = to_device(W_e)
    _, idx_base, idx_regroup, R1, R2, q1t_tau = _factor_and_select(Wd, params_r, tol_qr, null_pivots, tau=tau)
    numrank_W = len(idx_base)
    with single_threaded_blas():
        R1_inv = np.linalg.inv(R1)
        beta = np.around(np.dot(R1_inv, R2), 6)
        phi_b = np.round(np.dot(R1_inv, q1t_tau), 6)
    W_b = _base_columns(Wd, idx_base, on_dev)
    params_base = [params_r[i] for i in idx_base]
    params_regroup = [params_r[i] for i in idx_regroup]
    if params_std is not None:
        phi_std = [params_std[x] for x in params_base]
        for i in range(numrank_W):
            for j in range(beta.shape[1]):
                phi_std[i] = phi_std[i] + beta[i, j] * params_std[params_regroup[j]]
        phi_std = np.around(phi_std, 5)
    params_base = _expressions(params_base, params_regroup, beta)
    base_parameters = dict(zip(params_base, phi_b))
    if params_std is not None:
        return W_b, base_parameters, params_base, phi_b, phi_std
    return W_b, base_parameters, params_base, phi_b


@host_tail
def QR_pivoting(tau, W_e, params_r, tol_qr=TOL_QR):
    """(W_b, base_parameters) with column pivoting -- qrdecomposition.py:24-86.

    Pivoting decisions depend only on trailing column norms, which the orthogonal reduction
    preserves, so the pivoted factorisation is taken of the TSQR triangle.  The reference's rank
    loop leaves ``numrank_W = 0`` when no pivot falls below ``tol_qr``; kept as is.
    """
    from scipy import linalg

    Wd, on_dev = to_device(W_e)
    n = len(params_r)
    Raug = rfactor(Wd, tau=tau)
    Q2, R, P = linalg.qr(Raug[:n, :n], pivoting=True)
    params_rsorted = [params_r[P[i]] for i in range(P.shape[0])]
    numrank_W = 0
    diag = np.diag(R)
    for i in range(diag.shape[0]):
        if abs(diag[i]) > tol_qr:
            continue
        numrank_W = i
        break
    R1, R2 = R[:numrank_W, :numrank_W], R[:numrank_W, numrank_W:]
    R1_inv = np.linalg.inv(R1)
    beta = np.around(np.dot(R1_inv, R2), 6)
    phi_b = np.round(np.dot(R1_inv, np.dot(Q2[:, :numrank_W].T, Raug[:n, n])), 6)
    W_b = _base_columns(Wd, [int(P[i]) for i in range(numrank_W)], on_dev)
    params_base = _expressions(params_rsorted[:numrank_W], params_rsorted[numrank_W:], beta)
    return W_b, dict(zip(params_base, phi_b))


@host_tail
def cond_num(W_b, norm_type=None):
    """Condition number of the base regressor (qrdecomposition.py:316-332) from the singular values of
    its TSQR triangle (identical to those of ``W_b``)."""
    Wd, _ = to_device(W_b)
    if norm_type == "fro":
        if Wd.rows != Wd.cols:  # np.linalg.cond(.., 'fro') only takes square matrices
            raise np.linalg.LinAlgError("Last 2 dimensions of the array must be square")
        R = rfactor(Wd)
        return np.linalg.norm(R, "fro") * np.linalg.norm(np.linalg.inv(R), "fro")
    s = np.linalg.svd(rfactor(Wd), compute_uv=False)
    c = s.max() / s.min()
    if norm_type == "max_over_min_sigma":
        return c / (1.0 / c)
    return c
