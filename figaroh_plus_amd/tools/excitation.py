"""Condition-number objective of excitation-trajectory design.

The reference evaluates, thousands of times per Ipopt solve (examples/tiago/optimal_trajectory.py:43-70, 100-133),

    W   = build_regressor_basic(robot, q, v, a, param)
    W_b = build_baseRegressor(build_regressor_reduced(W, idx_e), idx_base)
    W_b = np.vstack((W_stack, W_b))          # optional: regressor of the trajectories found so far
    return np.linalg.cond(W_b)

The singular values of W_b are those of its R factor, and the R factor of a row-stacked matrix is the R factor of the
stacked triangles, so neither W nor W_b is ever stored: one streamed K1 -> TSQR pass over the trajectory samples
(``figh_regressor_tsqr``) gives an r x r triangle, ``figh_tsqr_merge`` folds in the triangle of the previous
trajectories, and the r x r SVD runs on the host.
"""
import numpy as np

from .. import _lib
from .._host import host_tail, singular_values_batch
from .regressor import _samples_to_device, regressor_flags


def base_columns(ncols, idx_e, idx_base):
    """Columns of W that make up W_b: build_regressor_reduced (np.delete) followed by build_baseRegressor."""
    gone = set(int(i) for i in idx_e)
    kept = [i for i in range(ncols) if i not in gone]
    return np.asarray([kept[int(i)] for i in idx_base], dtype=np.int32)


def base_regressor_triangle(robot, q, v, a, param, idx_e, idx_base, R_stack=None, coupling=False):
    """R factor (r x r, upper) of W_b for the samples (q, v, a), optionally of vstack((W_stack, W_b)) when the
    triangle ``R_stack`` of the previous trajectories is given."""
    mode, flags, ft_mask = regressor_flags(param, coupling)
    dm = robot.device_model()
    _, ncols = dm.shape(mode, flags)
    cols = base_columns(ncols, idx_e, idx_base)
    r = len(cols)
    N, d_q, d_v, d_a = _samples_to_device(robot.model, q, v, a)
    d_idx = _lib.DeviceArray.from_host(cols)
    d_R = _lib.DeviceArray((r * r,), np.float64)
    _lib.regressor_tsqr(dm, mode, flags, ft_mask, N, d_q, d_v, d_a, d_idx, r, None, None, d_R)
    if R_stack is not None:
        R_stack = np.ascontiguousarray(R_stack, dtype=np.float64)
        if R_stack.shape != (r, r):
            raise ValueError("R_stack must be the %d x %d triangle of the previous base regressor" % (r, r))
        pair = np.concatenate([R_stack.reshape(-1), d_R.to_host()])
        d_pair = _lib.DeviceArray.from_host(pair)
        _lib.tsqr_merge(d_pair, 2, r, d_R)
    return np.triu(d_R.to_host().reshape(r, r))


@host_tail
def objective_cond(robot, q, v, a, param, idx_e, idx_base, R_stack=None, coupling=False):
    """np.linalg.cond(W_b) of the reference's ``objective_func`` (2-norm condition number)."""
    R = base_regressor_triangle(robot, q, v, a, param, idx_e, idx_base, R_stack, coupling)
    s = np.linalg.svd(R, compute_uv=False)
    return float(s.max() / s.min())


def base_regressor_triangles_batch(robot, trajectories, param, idx_e, idx_base, R_stack=None, coupling=False):
    """R factors (B x r x r) of the base regressors of B trajectories ``[(q_b, v_b, a_b), ...]`` of equal length -- one
    K1 launch over all samples and one batched TSQR launch (``figh_regressor_tsqr_batch``) instead of B launch pairs."""
    if len(trajectories) == 0:
        raise ValueError("no trajectory given")
    mode, flags, ft_mask = regressor_flags(param, coupling)
    dm = robot.device_model()
    _, ncols = dm.shape(mode, flags)
    cols = base_columns(ncols, idx_e, idx_base)
    r = len(cols)
    n_per = len(trajectories[0][0])
    if n_per == 0 or any(len(t[0]) != n_per or len(t[1]) != n_per or len(t[2]) != n_per for t in trajectories):
        raise ValueError("the trajectories of a batch must have the same, non-zero number of samples")
    q = np.concatenate([np.asarray(t[0], dtype=np.float64) for t in trajectories])
    v = np.concatenate([np.asarray(t[1], dtype=np.float64) for t in trajectories])
    a = np.concatenate([np.asarray(t[2], dtype=np.float64) for t in trajectories])
    _, d_q, d_v, d_a = _samples_to_device(robot.model, q, v, a)
    B = len(trajectories)
    d_idx = _lib.DeviceArray.from_host(cols)
    d_stack = None
    if R_stack is not None:
        R_stack = np.ascontiguousarray(R_stack, dtype=np.float64)
        if R_stack.shape != (r, r):
            raise ValueError("R_stack must be the %d x %d triangle of the previous base regressor" % (r, r))
        d_stack = _lib.DeviceArray.from_host(np.triu(R_stack).reshape(-1))
    d_R = _lib.DeviceArray((B * r * r,), np.float64)
    _lib.regressor_tsqr_batch(dm, mode, flags, ft_mask, B, n_per, d_q, d_v, d_a, d_idx, r, d_stack, d_R)
    return np.triu(d_R.to_host().reshape(B, r, r))


@host_tail
def objective_cond_batch(robot, trajectories, param, idx_e, idx_base, R_stack=None, coupling=False):
    """``[np.linalg.cond(W_b) for every trajectory]``: the objective of examples/tiago/optimal_trajectory.py:100-133 at
    the B perturbed trajectories of one finite-difference gradient (numdifftools around ``objective_func``, :296-313),
    r x r SVDs batched on the host."""
    R = base_regressor_triangles_batch(robot, trajectories, param, idx_e, idx_base, R_stack, coupling)
    s = singular_values_batch(R)
    return (s.max(axis=1) / s.min(axis=1)).tolist()
