"""HIP-backed mirror of ``src/figaroh/tools/regressor.py:20-293``.

Same function names, argument order and return types as the reference; every
number is produced by the kernels in ``csrc/`` (no NumPy arithmetic on the
stacked regressor, no CPU fallback).  ``W`` arguments may be NumPy arrays or
:class:`figaroh_plus_amd.device.GpuMatrix`.
"""
import numpy as np

from .. import _lib
from .._host import host_tail
from ..device import GpuMatrix, index_to_device, to_device

FT_BITS = {"Fx": 0, "Fy": 1, "Fz": 2, "Mx": 3, "My": 4, "Mz": 5}


def regressor_flags(param, coupling=False):
    """param dict -> (mode, flags, ft_mask) of ``figh_regressor_build`` (regressor.py:45,55-70,89-140)."""
    flags = 0
    if param["has_friction"]:
        flags |= _lib.FLAG_FRICTION
    if param["has_actuator_inertia"]:
        flags |= _lib.FLAG_ACT_INERTIA
    if param["has_joint_offset"]:
        flags |= _lib.FLAG_OFFSET
    if coupling:
        flags |= _lib.FLAG_TX40
    if param.get("force_generic_kernel"):
        flags |= _lib.FLAG_GENERIC
    if param["is_joint_torques"]:
        return _lib.MODE_JOINT_TORQUE, flags, 63
    if param["is_external_wrench"]:
        ft_mask = 0
        for tok in param["force_torque"]:
            if tok == "All":
                ft_mask |= 63
            elif tok in FT_BITS:
                ft_mask |= 1 << FT_BITS[tok]
            else:
                raise ValueError("Please enter valid parameters")  # regressor.py:140
        return _lib.MODE_EXT_WRENCH, flags, ft_mask
    # the reference falls through both branches and fails on the unbound W_mod (regressor.py:194)
    raise UnboundLocalError("local variable 'W_mod' referenced before assignment")


def _samples_to_device(model, q, v, a):
    q = np.ascontiguousarray(q, dtype=np.float64)
    v = np.ascontiguousarray(v, dtype=np.float64)
    a = np.ascontiguousarray(a, dtype=np.float64)
    N = len(q)
    if q.shape != (N, model.nq) or v.shape != (N, model.nv) or a.shape != (N, model.nv):
        raise ValueError("q, v, a must have shapes (N, nq), (N, nv), (N, nv); got %r %r %r"
                         % (q.shape, v.shape, a.shape))
    return N, _lib.DeviceArray.from_host(q.reshape(-1)), _lib.DeviceArray.from_host(v.reshape(-1)), \
        _lib.DeviceArray.from_host(a.reshape(-1))


def build_regressor_device(robot, d_q, d_v, d_a, N, param, coupling=False, colsq=False, extra_flags=0):
    """Device-to-device core of :func:`build_regressor_basic`: returns (GpuMatrix W, DeviceArray colsq|None).
    ``extra_flags``: e.g. ``_lib.FLAG_BLOCKED_INPUTS`` when d_q / d_v / d_a are the tile-blocked copies."""
    mode, flags, ft_mask = regressor_flags(param, coupling)
    flags |= extra_flags
    handle = robot.device_model()
    rows_per_sample, ncols = handle.shape(mode, flags)
    W = GpuMatrix.empty(rows_per_sample * N, ncols)
    d_colsq = _lib.DeviceArray((ncols,), np.float64) if colsq else None
    _lib.regressor_build(handle, mode, flags, ft_mask, N, d_q, d_v, d_a, W.buf, ncols, d_colsq)
    return W, d_colsq


def build_regressor_basic(robot, q, v, a, param, tau=None):
    """Stacked regressor for the standard parameters (regressor.py:20-194).

    Returns a (N*nv, 14*nv) [joint torques] or (6N, 14*(njoints-1)) [external wrench] float64
    array in the reference's row / column order; ``tau`` is accepted and unused, as upstream.
    """
    N, d_q, d_v, d_a = _samples_to_device(robot.model, q, v, a)
    W, _ = build_regressor_device(robot, d_q, d_v, d_a, N, param)
    return W if param.get("device_resident") else W.numpy()


def add_coupling_TX40(W, model, data, N, nq, nv, njoints, q, v, a):
    """Append the [Iam6, fvm6, fsm6] columns of the Staubli TX40 wrist coupling (regressor.py:198-227)."""
    v = np.ascontiguousarray(v, dtype=np.float64)
    a = np.ascontiguousarray(a, dtype=np.float64)
    d_v, d_a = _lib.DeviceArray.from_host(v.reshape(-1)), _lib.DeviceArray.from_host(a.reshape(-1))
    extra = GpuMatrix.empty(6 * N, 3)
    _lib.coupling_tx40(N, v.shape[1], d_v, d_a, extra.buf)
    Wh = W.numpy() if isinstance(W, GpuMatrix) else np.asarray(W)
    if Wh.shape[0] != 6 * N:
        raise ValueError("W has %d rows, expected 6*N = %d" % (Wh.shape[0], 6 * N))
    return np.c_[Wh, extra.numpy()]


def _column_sqnorms(W):
    Wd, _ = to_device(W)
    out = _lib.DeviceArray((Wd.cols,), np.float64)
    _lib.colsq(Wd.buf, Wd.rows, Wd.cols, Wd.ld, out)
    return Wd, out.to_host()


def _split_columns(col_norm, params_std, tol_e):
    names = list(params_std.keys())
    idx_e, params_r = [], []
    for i in range(col_norm.shape[0]):
        if col_norm[i] < tol_e:
            idx_e.append(i)
        else:
            params_r.append(names[i])
    return idx_e, params_r


def _delete_columns(Wd, idx_e, keep_on_device):
    drop = set(int(i) for i in np.atleast_1d(np.asarray(idx_e, dtype=np.int64)))
    keep = [c for c in range(Wd.cols) if c not in drop]
    out = GpuMatrix.empty(Wd.rows, len(keep))
    if keep and Wd.rows:
        _lib.gather_cols(Wd.buf, Wd.rows, Wd.ld, index_to_device(keep), len(keep), out.buf, len(keep))
    return out if keep_on_device else out.numpy()


def eliminate_non_dynaffect(W, params_std, tol_e=1e-6):
    """Drop columns whose squared L2 norm is below ``tol_e`` (regressor.py:230-255): (W_e, params_r)."""
    Wd, col_norm = _column_sqnorms(W)
    idx_e, params_r = _split_columns(col_norm, params_std, tol_e)
    return _delete_columns(Wd, tuple(idx_e), isinstance(W, GpuMatrix)), params_r


def get_index_eliminate(W, params_std, tol_e=1e-6):
    """Indices of the columns to eliminate and names of the remaining ones (regressor.py:258-279)."""
    _, col_norm = _column_sqnorms(W)
    return _split_columns(col_norm, params_std, tol_e)


def build_regressor_reduced(W, idx_e):
    """``np.delete(W, idx_e, 1)`` on the device (regressor.py:282-293)."""
    Wd, on_dev = to_device(W)
    return _delete_columns(Wd, idx_e, on_dev)


# ---------------------------------------------------------------------------------------------------------------------
# Total-least-squares payload regressors (regressor.py:296-500), SURVEY section 8f-4.  W_tot is assembled in HBM block
# by block (figh_place_block: the concatenate / unary minus / zeros statements), its triangle comes from the TSQR
# kernel, the n x n SVD (n = columns of W_tot, a few dozen to a few hundred) runs on the host on that triangle -- the
# right singular vectors and singular values of W_tot are those of R -- and the residue is one device mat-vec.
def _place(src, r0, c0, rows, cols, scale, dst, dr0, dc0):
    _lib.place_block(src.ptr + 8 * (r0 * src.ld + c0), src.ld, rows, cols, scale,
                     dst.ptr + 8 * (dr0 * dst.ld + dc0), dst.ld)


def _blockdiag_columns(vec, nblocks, n, dst, r0, c0):
    """V_a / V_b of regressor.py:321-343 (and tau_ul / tau_l of :448-481): column ii holds entries ii*n .. (ii+1)*n - 1
    of ``vec`` in the same rows, zeros elsewhere."""
    d = GpuMatrix(_lib.DeviceArray.from_host(np.ascontiguousarray(vec, dtype=np.float64).reshape(-1)), nblocks * n, 1)
    for ii in range(nblocks):
        _place(d, ii * n, 0, n, 1, 1.0, dst, r0 + ii * n, c0 + ii)


def _tls_solution(Wt, mass_load):
    from .qrdecomposition import rfactor
    R = rfactor(Wt)
    _, _, Vh = np.linalg.svd(R, full_matrices=False)
    V = np.transpose(Vh).conj()
    V_norm = mass_load * np.divide(V[:, -1], V[-1, -1])
    d_x = _lib.DeviceArray.from_host(np.ascontiguousarray(V_norm))
    d_y = _lib.DeviceArray((Wt.rows,), np.float64)
    _lib.matvec(Wt.buf, Wt.rows, Wt.ld, None, Wt.cols, d_x, d_y)
    return V_norm, d_y.to_host()


def _payload_columns(W_l, first, picks, width, param_standard_l, eliminate):
    """W_l_temp of regressor.py:346-349 / :366-369 / :386-388 / :483-487: ``width`` columns, those in ``picks`` copied
    from W_l[:, first + k], the others zero; with ``eliminate`` the columns whose squared norm is below 1e-6 are dropped
    (get_index_eliminate + build_regressor_reduced)."""
    tmp = GpuMatrix.empty(W_l.rows, width)
    tmp.buf.zero_()
    if first + max(picks) >= W_l.cols:
        raise IndexError("index %d is out of bounds for axis 1 with size %d" % (first + max(picks), W_l.cols))
    for k in picks:
        _place(W_l, 0, first + k, W_l.rows, 1, 1.0, tmp, 0, k)
    if not eliminate:
        return tmp
    idx_e, _ = get_index_eliminate(tmp, param_standard_l, 1e-6)
    return build_regressor_reduced(tmp, idx_e)


def _total_regressor(W_b_u, W_b_l, W_l, meas_u, meas_l, nblocks, n_u, n_l, W_e_l, mass_col, param, on_dev):
    nb = W_b_u.cols
    if W_b_l.cols != nb:
        raise ValueError("all the input array dimensions except for the concatenation axis must match exactly")
    total = W_b_u.rows + W_b_l.rows
    if nblocks * (n_u + n_l) != total or 2 * W_l.rows != total:
        # np.concatenate((W_tot, W_current), axis=1) / ((W_tot, W_upayload), axis=1) of the reference: the measurement
        # columns have len(meas_u) + len(meas_l) rows, the payload columns 2 len(W_l)
        raise ValueError("all the input array dimensions except for the concatenation axis must match exactly")
    rows_u, rows_l = W_b_u.rows, W_b_l.rows
    ncols = nb + nblocks + W_e_l.cols + 1
    Wt = GpuMatrix.empty(rows_u + rows_l, ncols)
    Wt.buf.zero_()
    _place(W_b_u, 0, 0, rows_u, nb, -1.0, Wt, 0, 0)
    _place(W_b_l, 0, 0, rows_l, nb, -1.0, Wt, rows_u, 0)
    _blockdiag_columns(meas_u, nblocks, n_u, Wt, 0, nb)
    _blockdiag_columns(meas_l, nblocks, n_l, Wt, nblocks * n_u, nb)  # below the unloaded measurements
    _place(W_e_l, 0, 0, W_l.rows, W_e_l.cols, -1.0, Wt, W_l.rows, nb + nblocks)  # below len(W_l) zero rows
    _place(W_l, 0, mass_col, W_l.rows, 1, -1.0, Wt, W_l.rows, nb + nblocks + W_e_l.cols)
    V_norm, residue = _tls_solution(Wt, param["mass_load"])
    return (Wt if on_dev else Wt.numpy()), V_norm, residue


@host_tail
def build_total_regressor_current(W_b_u, W_b_l, W_l, I_u, I_l, param_standard_l, param):
    """(W_tot, V_norm, residue) -- regressor.py:296-412: unloaded and loaded base regressors stacked, the joint currents
    as block-diagonal columns (one drive gain per joint), the loaded link's inertial columns and its mass column; the
    total-least-squares solution is the right singular vector of the smallest singular value scaled to the known
    payload mass."""
    Wu, on_dev = to_device(W_b_u)
    Wl, _ = to_device(W_b_l)
    Wf, _ = to_device(W_l)
    n_samples = param["nb_samples"]
    nb_joints = int(len(I_u) / n_samples)
    if len(I_u) != nb_joints * n_samples or len(I_l) != len(I_u):
        raise ValueError("all the input array dimensions except for the concatenation axis must match exactly")
    body = param["which_body_loaded"]
    if param["has_friction"]:
        stride, picks = 12, [0, 1, 2, 3, 4, 5, 6, 7, 8, 10, 11]
    elif param["has_actuator_inertia"]:
        stride, picks = 14, [0, 1, 2, 3, 4, 5, 6, 7, 8, 10, 11, 12, 13]
    else:
        stride, picks = 10, list(range(9))
    width = stride if stride != 10 else 9
    W_e_l = _payload_columns(Wf, body * stride, picks, width, param_standard_l, True)
    if body * stride + 9 >= Wf.cols:
        raise IndexError("index %d is out of bounds for axis 1 with size %d" % (body * stride + 9, Wf.cols))
    return _total_regressor(Wu, Wl, Wf, I_u, I_l, nb_joints, n_samples, n_samples, W_e_l, body * stride + 9, param, on_dev)


@host_tail
def build_total_regressor_wrench(W_b_u, W_b_l, W_l, tau_u, tau_l, param_standard_l, param):
    """(W_tot, V_norm, residue) -- regressor.py:415-500: the external-wrench variant (six wrench components instead of
    joints, no column elimination on the payload block)."""
    Wu, on_dev = to_device(W_b_u)
    Wl, _ = to_device(W_b_l)
    Wf, _ = to_device(W_l)
    n_u, n_l = int(len(tau_u) / 6), int(len(tau_l) / 6)
    if len(tau_u) != 6 * n_u or len(tau_l) != 6 * n_l:
        raise ValueError("all the input array dimensions except for the concatenation axis must match exactly")
    body = param["which_body_loaded"]
    W_e_l = _payload_columns(Wf, body * 10, list(range(9)), 9, param_standard_l, False)
    if body * 10 + 9 >= Wf.cols:
        raise IndexError("index %d is out of bounds for axis 1 with size %d" % (body * 10 + 9, Wf.cols))
    return _total_regressor(Wu, Wl, Wf, tau_u, tau_l, 6, n_u, n_l, W_e_l, body * 10 + 9, param, on_dev)
