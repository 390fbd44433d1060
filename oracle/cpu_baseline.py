"""ORACLE-side CPU baseline (test infrastructure; used only by bench.py's cpu_baseline leg).

"Faithful" flavour of BASELINE.md section 3: keeps the REFERENCE'S STRUCTURE statement for statement --
a Python loop over samples around one native per-sample regressor call (Pinocchio's C++ kernel in the
reference, oracle/figh_oracle.c here, reached through ctypes), the per-row NumPy scatter and the 14*nv
strided column copies (src/figaroh/tools/regressor.py:48-87), ``np.diag(np.dot(W.T, W))`` (:271),
``np.delete`` (:292), two ``np.linalg.qr`` with explicit Q and ``inv(R1) @ R2``
(src/figaroh/tools/qrdecomposition.py:205-244) and ``np.linalg.pinv(W_b) @ tau``
(examples/ur10/identification.py:159).  This is what a FIGAROH user runs, with Pinocchio's inner
kernel replaced by the C restatement.  kind = "port".
"""
import time

import numpy as np

import oracle_c


def faithful_pass(flat, q, v, a, tau=None, tol_e=1e-6, tol_qr=1e-8, friction=False, actuator_inertia=False, offset=False):
    """One pass of the hot path on the CPU for a joint-torque model (BASELINE configs 2 and 3; the friction /
    actuator-inertia / offset columns of regressor.py:55-70 when the flags are set).  Returns (result dict, seconds per
    stage)."""
    om = oracle_c.OracleModel(flat)
    N, nv = len(q), om.nv
    t = {}
    t0 = time.perf_counter()
    W = np.zeros([N * nv, 14 * nv])
    W_mod = np.zeros([N * nv, 14 * nv])
    Y = np.empty((nv, 10 * nv))
    for i in range(N):
        W_temp = om.joint_torque_regressor(q[i, :], v[i, :], a[i, :], out=Y)
        for j in range(W_temp.shape[0]):
            W[j * N + i, 0:10 * nv] = W_temp[j, :]
            if friction:
                W[j * N + i, 10 * nv + 2 * j] = v[i, j]
                W[j * N + i, 10 * nv + 2 * j + 1] = np.sign(v[i, j])
            else:
                W[j * N + i, 10 * nv + 2 * j] = 0
                W[j * N + i, 10 * nv + 2 * j + 1] = 0
            if actuator_inertia:
                W[j * N + i, 10 * nv + 2 * nv + j] = a[i, j]
            else:
                W[j * N + i, 10 * nv + 2 * nv + j] = 0
            if offset:
                W[j * N + i, 10 * nv + 2 * nv + nv + j] = 1
            else:
                W[j * N + i, 10 * nv + 2 * nv + nv + j] = 0
    src = [4, 5, 7, 6, 8, 9, 1, 2, 3, 0]
    for k in range(nv):
        for dst in range(10):
            W_mod[:, 14 * k + dst] = W[:, 10 * k + src[dst]]
        W_mod[:, 14 * k + 10] = W[:, 10 * nv + 2 * nv + k]
        W_mod[:, 14 * k + 11] = W[:, 10 * nv + 2 * k]
        W_mod[:, 14 * k + 12] = W[:, 10 * nv + 2 * k + 1]
        W_mod[:, 14 * k + 13] = W[:, 10 * nv + 2 * nv + nv + k]
    t["regressor"] = time.perf_counter() - t0
    t0 = time.perf_counter()
    col_norm = np.diag(np.dot(W_mod.T, W_mod))
    idx_e = [i for i in range(col_norm.shape[0]) if col_norm[i] < tol_e]
    W_e = np.delete(W_mod, idx_e, 1)
    t["eliminate"] = time.perf_counter() - t0
    t0 = time.perf_counter()
    Q, R = np.linalg.qr(W_e)
    d = np.abs(np.diag(R))
    idx_base = [i for i in range(len(d)) if d[i] > tol_qr]
    idx_regroup = [i for i in range(len(d)) if not d[i] > tol_qr]
    W1, W2 = W_e[:, idx_base], W_e[:, idx_regroup]
    Q_r, R_r = np.linalg.qr(np.c_[W1, W2])
    r = len(idx_base)
    beta = np.around(np.matmul(np.linalg.inv(R_r[:r, :r]), R_r[:r, r:]), 6)
    W_b = np.dot(Q_r[:, :r], R_r[:r, :r])
    assert np.allclose(W1, W_b)
    t["base_qr"] = time.perf_counter() - t0
    out = {"idx_e": idx_e, "idx_base": idx_base, "beta": beta}
    if tau is not None:
        t0 = time.perf_counter()
        out["phi"] = np.matmul(np.linalg.pinv(W_b), tau)
        t["pinv"] = time.perf_counter() - t0
    return out, t


def fast_pass(flat, q, v, a, tau=None, tol_e=1e-6, tol_qr=1e-8, flags=0):
    """"Fair-fast" flavour: the same results with the CPU used well -- one native call for the
    whole batch (oracle/figh_oracle.c, OpenMP over the samples, W written once in its final layout), column norms
    without the Gram product, ``np.linalg.qr(mode='r')`` of [W_e tau] (no Q), the regrouped factorisation on the
    n x n triangle, triangular solves.  Returns (result dict, seconds per stage)."""
    om = oracle_c.OracleModel(flat)
    t = {}
    t0 = time.perf_counter()
    W = om.build_regressor_basic(q, v, a, 0, flags)
    t["regressor"] = time.perf_counter() - t0
    t0 = time.perf_counter()
    col_norm = np.einsum("ij,ij->j", W, W)
    kept = [i for i in range(W.shape[1]) if not col_norm[i] < tol_e]
    idx_e = [i for i in range(W.shape[1]) if col_norm[i] < tol_e]
    A = W[:, kept] if tau is None else np.c_[W[:, kept], tau]
    t["eliminate"] = time.perf_counter() - t0
    t0 = time.perf_counter()
    R = np.linalg.qr(A, mode="r")
    n = len(kept)
    d = np.abs(np.diag(R))[:n]
    idx_base = [i for i in range(n) if d[i] > tol_qr]
    idx_regroup = [i for i in range(n) if not d[i] > tol_qr]
    perm = idx_base + idx_regroup + ([n] if tau is not None else [])
    R2 = np.linalg.qr(R[:, perm], mode="r")
    r = len(idx_base)
    import scipy.linalg as sl
    beta = np.around(sl.solve_triangular(R2[:r, :r], R2[:r, r:n]), 6)
    t["base_qr"] = time.perf_counter() - t0
    out = {"idx_e": idx_e, "idx_base": idx_base, "beta": beta}
    if tau is not None:
        t0 = time.perf_counter()
        out["phi"] = sl.solve_triangular(R2[:r, :r], R2[:r, n])
        t["solve"] = time.perf_counter() - t0
    return out, t
