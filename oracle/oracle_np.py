"""ORACLE (test infrastructure, NOT product code) -- NumPy restatement of the reference path.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this module; nothing under ``figaroh_plus_amd/`` does.

What is restated, and what pins it
----------------------------------
* ``joint_torque_regressor`` restates ``pinocchio.computeJointTorqueRegressor``
  -- third-party, un-vendored, version un-pinned in the reference
  (``environment.yml:7``); called at ``src/figaroh/tools/regressor.py:49-51`` and
  ``:93-95``.  Pinocchio is not installable here, so for the floating-point
  entries of the per-sample regressor **parity is unpinned by the reference
  itself**.  The restatement follows Pinocchio's published algorithm (forward
  pass of spatial velocities / accelerations with a_0 = -gravity, backward pass
  of the 6x10 body regressor up the parent chain) and is
  pinned instead by (i) an independent RNEA (``rnea`` below): W.phi_urdf == tau
  to ~1e-13, and (ii) the reference's committed known-answer artefacts reached
  *through the reference's own code* (``oracle/gen_golden.py`` imports
  ``regressor.py`` / ``qrdecomposition.py`` / ``robot.py`` from /root/reference
  on top of this function): the 60 base-parameter expressions of
  ``examples/staubli_TX40/results/TX40_bp_5.csv``, the TIAGo joint numbering and
  the human body list, and (iii) the NUMBERS of that CSV (phi_OLS, sigma%, phi_WLS),
  reproduced from the reference's committed TX40 measurements to <= 3.8e-4
  (``oracle/gen_golden_tx40_real.py``, ``tests/test_oracle.py::test_tx40_real_data_known_answers``), and (iv) for
  fixed-base TREES the reference's committed, Pinocchio-produced TIAGo result
  ``examples/tiago/data/identification/dynamic/tiago_bp_19_Oct_2024_2320.csv``: the TIAGo script replayed on the
  committed TIAGo measurements reproduces its 44 expressions verbatim and its values / sigma% exactly at the
  file's precision (``oracle/gen_golden_tiago_real.py``, ``test_tiago_real_data_known_answers``).  What no Pinocchio
  output pins is the free-flyer frame convention (the reference commits no floating-base result).
* everything else (row/column layout, elimination, QR bookkeeping, strings,
  LS/WLS/sigma) restates plain NumPy code of the reference and is pinned against
  outputs of the reference itself (``tests/golden/*.npz``).

All 6-vectors are (linear, angular).  Placements map child coords to parent coords.
"""
from __future__ import annotations

import numpy as np

JT_REVOLUTE, JT_PRISMATIC, JT_CONTINUOUS, JT_FREEFLYER = 0, 1, 2, 3

# Pinocchio slot [m mx my mz Ixx Ixy Iyy Ixz Iyz Izz] -> FIGAROH slot inside a 14-wide link block
# (src/figaroh/tools/regressor.py:72-82 and :171-182)
PIN_TO_FIG = np.array([9, 6, 7, 8, 0, 1, 3, 2, 4, 5])
PARAM_NAMES = ("Ixx", "Ixy", "Ixz", "Iyy", "Iyz", "Izz", "mx", "my", "mz", "m")
FT_ROWS = {"Fx": 0, "Fy": 1, "Fz": 2, "Mx": 3, "My": 4, "Mz": 5}


def skew(x):
    return np.array([[0.0, -x[2], x[1]], [x[2], 0.0, -x[0]], [-x[1], x[0], 0.0]])


def rodrigues(axis, c, s):
    K = skew(axis)
    return np.eye(3) + s * K + (1.0 - c) * (K @ K)


def _joint_transform(flat, i, q):
    """(R, p) of joint i's own motion and its 6 x nv_i motion subspace."""
    jt = int(flat["jtype"][i])
    ax = np.asarray(flat["axis"][i], dtype=float)
    iq = int(flat["idx_q"][i])
    if jt == JT_REVOLUTE:
        return rodrigues(ax, np.cos(q[iq]), np.sin(q[iq])), np.zeros(3), np.concatenate([np.zeros(3), ax])[:, None]
    if jt == JT_CONTINUOUS:
        return rodrigues(ax, q[iq], q[iq + 1]), np.zeros(3), np.concatenate([np.zeros(3), ax])[:, None]
    if jt == JT_PRISMATIC:
        return np.eye(3), ax * q[iq], np.concatenate([ax, np.zeros(3)])[:, None]
    if jt == JT_FREEFLYER:
        x, y, z, w = q[iq + 3:iq + 7]
        R = np.array([
            [1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
            [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
            [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)],
        ])
        return R, np.array(q[iq:iq + 3], dtype=float), np.eye(6)
    raise ValueError("joint type %d" % jt)


def _nv_of(flat, i):
    return 6 if int(flat["jtype"][i]) == JT_FREEFLYER else 1


def forward_pass(flat, q, v, a):
    n = int(flat["njoints"])
    liMi = [None] * n
    S = [None] * n
    V = [np.zeros(6) for _ in range(n)]
    A = [np.zeros(6) for _ in range(n)]
    A[0] = np.concatenate([-np.asarray(flat["gravity"], dtype=float), np.zeros(3)])
    for i in range(1, n):
        Rj, pj, Si = _joint_transform(flat, i, q)
        Rp = np.asarray(flat["placement"][i][:9], dtype=float).reshape(3, 3)
        pp = np.asarray(flat["placement"][i][9:], dtype=float)
        R, p = Rp @ Rj, Rp @ pj + pp
        liMi[i], S[i] = (R, p), Si
        iv, nvi = int(flat["idx_v"][i]), _nv_of(flat, i)
        par = int(flat["parents"][i])

        def to_child(m):
            return np.concatenate([R.T @ (m[:3] - np.cross(p, m[3:])), R.T @ m[3:]])

        vj = Si @ v[iv:iv + nvi]
        Vi = vj + to_child(V[par])
        cross = np.concatenate([np.cross(Vi[3:], vj[:3]) + np.cross(Vi[:3], vj[3:]),
                                np.cross(Vi[3:], vj[3:])])
        V[i] = Vi
        A[i] = cross + Si @ a[iv:iv + nvi] + to_child(A[par])
    return liMi, S, V, A


def body_regressor(V, A):
    """6x10 matrix B with  f_body = B . [m, m c, Ixx Ixy Iyy Ixz Iyz Izz]."""
    vl, w = V[:3], V[3:]
    al, dw = A[:3], A[3:]
    acc = al + np.cross(w, vl)

    def L(x):
        return np.array([[x[0], x[1], 0, x[2], 0, 0],
                         [0, x[0], x[1], 0, x[2], 0],
                         [0, 0, 0, x[0], x[1], x[2]]], dtype=float)

    B = np.zeros((6, 10))
    B[:3, 0] = acc
    B[:3, 1:4] = skew(dw) + skew(w) @ skew(w)
    B[3:, 1:4] = -skew(acc)
    B[3:, 4:] = L(dw) + skew(w) @ L(w)
    return B


def joint_torque_regressor(flat, q, v, a):
    """Restatement of pinocchio.computeJointTorqueRegressor: (nv, 10*(njoints-1))."""
    n = int(flat["njoints"])
    liMi, S, V, A = forward_pass(flat, q, v, a)
    Y = np.zeros((int(flat["nv"]), 10 * (n - 1)))
    for i in range(n - 1, 0, -1):
        B = body_regressor(V[i], A[i])
        j = i
        while j > 0:
            iv, nvj = int(flat["idx_v"][j]), _nv_of(flat, j)
            Y[iv:iv + nvj, 10 * (i - 1):10 * i] = S[j].T @ B
            R, p = liMi[j]
            lin = R @ B[:3]
            B = np.vstack([lin, R @ B[3:] + skew(p) @ lin])
            j = int(flat["parents"][j])
    return Y


def spatial_inertia(mass, lever, inertia_com):
    s = skew(lever)
    Io = inertia_com + mass * (s.T @ s)
    M = np.zeros((6, 6))
    M[:3, :3] = mass * np.eye(3)
    M[:3, 3:] = -mass * s
    M[3:, :3] = mass * s
    M[3:, 3:] = Io
    return M


def rnea(flat, q, v, a):
    """Independent recursive Newton-Euler (physics cross-check for the regressor)."""
    n = int(flat["njoints"])
    liMi, S, V, A = forward_pass(flat, q, v, a)
    f = [np.zeros(6) for _ in range(n)]
    for i in range(1, n):
        I6 = spatial_inertia(float(flat["mass"][i]), np.asarray(flat["lever"][i], dtype=float),
                             np.asarray(flat["inertia"][i], dtype=float).reshape(3, 3))
        h = I6 @ V[i]
        vl, w = V[i][:3], V[i][3:]
        f[i] = I6 @ A[i] + np.concatenate([np.cross(w, h[:3]), np.cross(w, h[3:]) + np.cross(vl, h[:3])])
    tau = np.zeros(int(flat["nv"]))
    for i in range(n - 1, 0, -1):
        iv, nvi = int(flat["idx_v"][i]), _nv_of(flat, i)
        tau[iv:iv + nvi] = S[i].T @ f[i]
        R, p = liMi[i]
        lin = R @ f[i][:3]
        par = int(flat["parents"][i])
        f[par] = f[par] + np.concatenate([lin, R @ f[i][3:] + np.cross(p, lin)])
    return tau


def rnea_with_parameters(flat, q, v, a, pi):
    """Recursive Newton-Euler with the inertial parameters given explicitly: ``pi`` is (njoints-1, 10) in Pinocchio's order
    [m, m c, I_O(xx xy yy xz yz zz)] (first moment and inertia about the joint origin).  tau is linear in pi, so
    ``rnea_with_parameters(.., e_(i,p))`` is column 10 (i-1) + p of the joint-torque regressor: the column-by-column pin of
    ``joint_torque_regressor`` (tests/test_oracle.py).  Nothing here touches ``body_regressor``."""
    n = int(flat["njoints"])
    liMi, S, V, A = forward_pass(flat, q, v, a)
    f = [np.zeros(6) for _ in range(n)]
    for i in range(1, n):
        m, h = pi[i - 1][0], np.asarray(pi[i - 1][1:4], dtype=float)
        xx, xy, yy, xz, yz, zz = pi[i - 1][4:]
        Io = np.array([[xx, xy, xz], [xy, yy, yz], [xz, yz, zz]])
        I6 = np.zeros((6, 6))
        I6[:3, :3] = m * np.eye(3)
        I6[:3, 3:] = -skew(h)
        I6[3:, :3] = skew(h)
        I6[3:, 3:] = Io
        mom = I6 @ V[i]
        vl, w = V[i][:3], V[i][3:]
        f[i] = I6 @ A[i] + np.concatenate([np.cross(w, mom[:3]), np.cross(w, mom[3:]) + np.cross(vl, mom[:3])])
    tau = np.zeros(int(flat["nv"]))
    for i in range(n - 1, 0, -1):
        iv, nvi = int(flat["idx_v"][i]), _nv_of(flat, i)
        tau[iv:iv + nvi] = S[i].T @ f[i]
        R, p = liMi[i]
        lin = R @ f[i][:3]
        par = int(flat["parents"][i])
        f[par] = f[par] + np.concatenate([lin, R @ f[i][3:] + np.cross(p, lin)])
    return tau


def dynamic_parameters(flat):
    """Pinocchio-ordered [m, mc, Ixx Ixy Iyy Ixz Iyz Izz] (about the joint origin), joints 1..n-1."""
    out = []
    for i in range(1, int(flat["njoints"])):
        m = float(flat["mass"][i])
        c = np.asarray(flat["lever"][i], dtype=float)
        s = skew(c)
        Io = np.asarray(flat["inertia"][i], dtype=float).reshape(3, 3) + m * (s.T @ s)
        out.append([m, m * c[0], m * c[1], m * c[2], Io[0, 0], Io[0, 1], Io[1, 1], Io[0, 2], Io[1, 2], Io[2, 2]])
    return np.array(out)


# ----------------------------------------------------------------------------- layout
def ft_rows(force_torque):
    """Rows of the 6-row external-wrench block that get inertial columns
    (src/figaroh/tools/regressor.py:96-140)."""
    rows = set()
    for tok in force_torque:
        if tok == "All":
            rows.update(range(6))
        elif tok in FT_ROWS:
            rows.add(FT_ROWS[tok])
        else:
            raise ValueError("Please enter valid parameters")
    return sorted(rows)


def build_regressor_basic(flat, q, v, a, param):
    """Stacked regressor in the reference's layout (src/figaroh/tools/regressor.py:20-194).

    joint-torque mode: rows r = j*N + i, columns 14*k + slot for link k (needs njoints-1 == nv);
    external-wrench mode: rows r = c*N + i for wrench component c, inertial columns only for
    bodies with mass != 0 and components named in param['force_torque']; fv/fs/Ia/off columns
    use v[i, k], a[i, k] (k = link index) on *all six* rows.
    """
    q, v, a = (np.asarray(x, dtype=float) for x in (q, v, a))
    N = len(q)
    nl = int(flat["njoints"]) - 1
    if param["is_joint_torques"]:
        nv = int(flat["nv"])
        W = np.zeros((N * nv, 14 * nv))
        for i in range(N):
            Y = joint_torque_regressor(flat, q[i], v[i], a[i])
            for j in range(nv):
                r = j * N + i
                for k in range(nv):
                    W[r, 14 * k + PIN_TO_FIG] = Y[j, 10 * k:10 * k + 10]
                if param["has_actuator_inertia"]:
                    W[r, 14 * j + 10] = a[i, j]
                if param["has_friction"]:
                    W[r, 14 * j + 11] = v[i, j]
                    W[r, 14 * j + 12] = np.sign(v[i, j])
                if param["has_joint_offset"]:
                    W[r, 14 * j + 13] = 1.0
        return W
    if param["is_external_wrench"]:
        rows = ft_rows(param["force_torque"])
        bodies = [k for k in range(1, nl + 1) if float(flat["mass"][k]) != 0.0]
        W = np.zeros((N * 6, 14 * nl))
        for i in range(N):
            Y = joint_torque_regressor(flat, q[i], v[i], a[i])
            for c in rows:
                for k in bodies:
                    W[c * N + i, 14 * (k - 1) + PIN_TO_FIG] = Y[c, 10 * (k - 1):10 * k]
            for k in range(nl):
                for c in range(6):
                    r = c * N + i
                    if param["has_actuator_inertia"]:
                        W[r, 14 * k + 10] = a[i, k]
                    if param["has_friction"]:
                        W[r, 14 * k + 11] = v[i, k]
                        W[r, 14 * k + 12] = np.sign(v[i, k])
                    if param["has_joint_offset"]:
                        W[r, 14 * k + 13] = 1.0
        return W
    return None  # the reference raises UnboundLocalError here (regressor.py:194)


def add_coupling_TX40(W, N, v, a):
    """src/figaroh/tools/regressor.py:198-227."""
    W = np.c_[W, np.zeros((W.shape[0], 3))]
    i = np.arange(N)
    s = np.sign(v[:, 4] + v[:, 5])
    W[4 * N + i, -3], W[4 * N + i, -2], W[4 * N + i, -1] = a[:, 5], v[:, 5], s
    W[5 * N + i, -3], W[5 * N + i, -2], W[5 * N + i, -1] = a[:, 4], v[:, 4], s
    return W


def standard_parameter_names(nl, coupling=False):
    """Key order of Robot.get_standard_parameters (src/figaroh/tools/robot.py:102-155)."""
    names = []
    for i in range(1, nl + 1):
        names += [p + str(i) for p in PARAM_NAMES]
        names += ["Ia%d" % i, "fv%d" % i, "fs%d" % i, "off%d" % i]
    if coupling:
        names += ["Iam6", "fvm6", "fsm6"]
    return names


def standard_parameters(flat, param, coupling=False):
    """Ordered name -> value dict (src/figaroh/tools/robot.py:76-155, TX40 extras
    examples/staubli_TX40/identification.py:61-64)."""
    P = dynamic_parameters(flat)
    nl = int(flat["njoints"]) - 1
    vals = []

    def pick(key, i, n=1):
        try:
            return [param[key][i]] if n == 1 else None
        except Exception:
            return [0]

    for i in range(nl):
        blk = np.zeros(10)
        blk[PIN_TO_FIG] = P[i]
        vals += list(blk)
        vals += pick("Ia", i) if param["has_actuator_inertia"] else [0]
        if param["has_friction"]:
            try:
                vals += [param["fv"][i], param["fs"][i]]
            except Exception:
                vals += [0, 0]
        else:
            vals += [0, 0]
        vals += pick("off", i) if param["has_joint_offset"] else [0]
    names = standard_parameter_names(nl, False)
    d = dict(zip(names, vals))
    if coupling:
        d["Iam6"], d["fvm6"], d["fsm6"] = param["Iam6"], param["fvm6"], param["fsm6"]
    return d


# ----------------------------------------------------------------------------- elimination + QR
def get_index_eliminate(W, names, tol_e=1e-6):
    """src/figaroh/tools/regressor.py:258-279 (diag(W^T W) < tol)."""
    col = np.einsum("ij,ij->j", W, W)
    idx_e = [i for i in range(W.shape[1]) if col[i] < tol_e]
    params_r = [names[i] for i in range(W.shape[1]) if not col[i] < tol_e]
    return idx_e, params_r


def regroup_strings(params_base, params_regroup, beta, tol_beta=1e-6):
    """String building of src/figaroh/tools/qrdecomposition.py:246-266."""
    out = list(params_base)
    for i in range(beta.shape[0]):
        for j in range(beta.shape[1]):
            if abs(beta[i, j]) < tol_beta:
                continue
            sign = " - " if beta[i, j] < -tol_beta else " + "
            out[i] = out[i] + sign + str(abs(beta[i, j])) + "*" + str(params_regroup[j])
    return out


def base_parameters(W_e, params_r, tol_qr=1e-8, tau=None):
    """Selection / regrouping of qrdecomposition.py:190-271 (and :89-187 when tau is given).

    Returns dict(idx_base, beta, params_base, W_b[, phi_b]).
    """
    R = np.linalg.qr(W_e, mode="r")
    assert R.shape[0] == len(params_r), "params_r does not have same length with R"
    d = np.abs(np.diag(R))
    idx_base = [i for i in range(len(params_r)) if d[i] > tol_qr]
    idx_regroup = [i for i in range(len(params_r)) if not d[i] > tol_qr]
    r = len(idx_base)
    Wr = np.c_[W_e[:, idx_base], W_e[:, idx_regroup]]
    Q_r, R_r = np.linalg.qr(Wr)
    R1, R2 = R_r[:r, :r], R_r[:r, r:]
    beta = np.around(np.linalg.inv(R1) @ R2, 6)
    out = {
        "idx_base": idx_base,
        "beta": beta,
        "params_base": regroup_strings([params_r[i] for i in idx_base], [params_r[i] for i in idx_regroup], beta),
        "W_b": W_e[:, idx_base].copy(),
        "diagR": np.diag(R).copy(),
    }
    if tau is not None:
        out["phi_b"] = np.round(np.linalg.inv(R1) @ (Q_r[:, :r].T @ tau), 6)
    return out


def relative_stdev(W_b, phi_b, tau):
    """src/figaroh/identification/identification_tools.py:204-234."""
    phi_b = np.asarray(phi_b, dtype=float)
    sig2 = np.linalg.norm(tau - W_b @ phi_b) ** 2 / (W_b.shape[0] - phi_b.shape[0])
    C = sig2 * np.linalg.inv(W_b.T @ W_b)
    with np.errstate(divide="ignore", invalid="ignore"):  # (phi_i == 0: inf, as the reference's scalar division gives)
        return np.round(100 * np.sqrt(np.diag(C)) / np.abs(phi_b), 2)


def wls_script(W_b, tau, phi_b, row_counts):
    """Script WLS of examples/staubli_TX40/identification.py:305-346 without the dense SIGMA."""
    phi_b = np.asarray(phi_b, dtype=float)
    w = np.zeros(W_b.shape[0])
    a = 0
    for n in row_counts:
        res = tau[a:a + n] - W_b[a:a + n] @ phi_b
        w[a:a + n] = 1.0 / (np.linalg.norm(res) ** 2 / n)
        a += n
    C = np.linalg.inv(W_b.T @ (W_b * w[:, None]))
    phi = np.around(C @ (W_b.T @ (w * tau)), 6)
    with np.errstate(divide="ignore", invalid="ignore"):  # (phi_i == 0: inf, as the script's scalar division gives)
        std = np.round(100 * np.sqrt(np.diag(C)) / np.abs(phi), 2)
    return phi, std


def essential_script(W_b, tau, params_base, std_xr, sig_ro_joint, row_counts, ratio_essential):
    """The essential-parameter loop of examples/staubli_TX40/identification.py:354-399, statement for statement (dense
    SIGMA replaced by its diagonal): while max(std) >= ratio * min(std) drop the parameter with the largest std%, OLS
    (lstsq, 6 decimals) + relative_stdev, WLS with the variances ``sig_ro_joint`` of the FULL base set (the script fills
    diag_SIGMA_e from sig_ro_joint, not from sig_ro_joint_e), std% from C_X_e."""
    names = list(params_base)
    W_ess = np.array(W_b, dtype=float)
    std_e = np.asarray(std_xr, dtype=float)
    w = np.repeat(1.0 / np.asarray(sig_ro_joint, dtype=float), row_counts)
    out = {"iterations": 0}
    while not (std_e.max() < ratio_essential * std_e.min()):
        (i,) = np.where(np.isclose(std_e, std_e.max()))
        del names[int(i[0])]
        W_ess = np.delete(W_ess, i, 1)
        phi_e_ols = np.around(np.linalg.lstsq(W_ess, tau, rcond=None)[0], 6)
        std_e_ols = relative_stdev(W_ess, phi_e_ols, tau)
        C = np.linalg.inv(W_ess.T @ (W_ess * w[:, None]))
        phi_e_wls = np.around(C @ (W_ess.T @ (w * tau)), 6)
        with np.errstate(divide="ignore", invalid="ignore"):
            std_e = np.round(100 * np.sqrt(np.diag(C)) / np.abs(phi_e_wls), 2)
        out.update(phi_e_ols=phi_e_ols, std_e_ols=std_e_ols, phi_e_wls=phi_e_wls, std_e_wls=std_e.copy())
        out["iterations"] += 1
    out["params_essential"] = names
    return out


def weigthed_least_squares(nq, phi_b, W_b, tau_meas, tau_est, idx_tau_stop):
    """Library WLS of src/figaroh/identification/identification_tools.py:291-331,
    including its quirks (sigma is a norm, not a variance; re-solve inside the joint loop)."""
    phi_b = np.asarray(phi_b, dtype=float)
    p = np.zeros(len(tau_meas))
    nb = int(idx_tau_stop[0])
    start = 0
    for ii in range(nq):
        stop = int(idx_tau_stop[ii])
        diff = tau_meas[start:stop] - tau_est[start:stop]
        sigma = np.linalg.norm(diff) / (len(diff) - len(phi_b))
        start = stop
        p[ii * nb:(ii + 1) * nb] = 1.0 / sigma
        phi_b = np.linalg.pinv(W_b * p[:, None]) @ (p * tau_meas)
    return np.around(phi_b, 6)


# ------------------------------------------------------------------------------------------------ preprocessing
def low_pass_filter_data(data, param, nbutter=5):
    """ORACLE: zero-phase Butterworth low-pass + border trimming (identification_tools.py:390-424), SciPy on the host."""
    from scipy import signal

    cutoff = param["ts"] * param["cut_off_frequency_butterworth"] / 2
    b, a = signal.butter(nbutter, cutoff, "low")
    padlen = 3 * (max(len(b), len(a)) - 1)
    data = signal.filtfilt(b, a, data, axis=0, padtype="odd", padlen=padlen)
    nbord = 5 * nbutter
    return data[nbord:data.shape[0] - nbord]


def decimate_joint_blocks(W, tau, nblocks, q=10, stages=2):
    """ORACLE: per-joint ``scipy.signal.decimate(zero_phase=True)`` of tau and of every column of W
    (examples/staubli_TX40/identification.py:186-204, examples/tiago/identification.py:142-187)."""
    from scipy import signal

    W = np.asarray(W)
    nj = tau.shape[0] // nblocks
    W_list, tau_list = [], []
    for i in range(nblocks):
        t = tau[i * nj:(i + 1) * nj]
        blk = W[i * nj:i * nj + nj]
        for _ in range(stages):
            t = signal.decimate(t, q=q, zero_phase=True)
            blk = signal.decimate(blk, q=q, zero_phase=True, axis=0)
        W_list.append(np.ascontiguousarray(blk))
        tau_list.append(t)
    return W_list, tau_list


# ------------------------------------------------------------------------------------------- pin.difference (restated)
def quat_to_rot(qx, qy, qz, qw):
    x, y, z, w = qx, qy, qz, qw
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def log3(R):
    """Rotation vector of R (Pinocchio log3): angle from the trace, axis from the antisymmetric part."""
    tr = np.clip((np.trace(R) - 1.0) / 2.0, -1.0, 1.0)
    theta = np.arccos(tr)
    w = np.array([R[2, 1] - R[1, 2], R[0, 2] - R[2, 0], R[1, 0] - R[0, 1]])
    if theta < 1e-8:
        return 0.5 * w
    if np.pi - theta < 1e-6:  # near pi: axis from the symmetric part
        A = (R + np.eye(3)) / 2.0
        k = int(np.argmax(np.diag(A)))
        ax = A[:, k] / np.sqrt(A[k, k])
        if w @ ax < 0:
            ax = -ax
        return theta * ax
    return theta / (2.0 * np.sin(theta)) * w


def log6(R, p):
    """Twist (v, w) with exp6(v, w) = (R, p) -- Pinocchio log6 (Spatial/explog.hpp): w = log3(R),
    v = alpha p - w x p / 2 + beta (w . p) w."""
    w = log3(R)
    t = np.linalg.norm(w)
    if t < 1e-4:
        alpha = 1.0 - t * t / 12.0 - t ** 4 / 720.0
        beta = 1.0 / 12.0 + t * t / 720.0
    else:
        st, ct = np.sin(t), np.cos(t)
        alpha = t * st / (2.0 * (1.0 - ct))
        beta = 1.0 / (t * t) - st / (2.0 * t * (1.0 - ct))
    v = alpha * p - 0.5 * np.cross(w, p) + beta * (w @ p) * w
    return v, w


def joint_difference(flat, q0, q1):
    """pin.difference(model, q0, q1) (call sites: identification_tools.py:370,376): the tangent vector that takes q0 to
    q1.  Revolute / prismatic: q1 - q0; continuous (cos, sin): the angle of R0^T R1; free-flyer: log6(M0^-1 M1) in the
    local frame of M0, ordered (linear, angular)."""
    out = np.zeros(int(flat["nv"]))
    for j in range(1, int(flat["njoints"])):
        jt, iq, iv = int(flat["jtype"][j]), int(flat["idx_q"][j]), int(flat["idx_v"][j])
        if jt in (0, 1):
            out[iv] = q1[iq] - q0[iq]
        elif jt == 2:
            c0, s0, c1, s1 = q0[iq], q0[iq + 1], q1[iq], q1[iq + 1]
            out[iv] = np.arctan2(s1 * c0 - c1 * s0, c1 * c0 + s1 * s0)
        else:
            R0, R1 = quat_to_rot(*q0[iq + 3:iq + 7]), quat_to_rot(*q1[iq + 3:iq + 7])
            v, w = log6(R0.T @ R1, R0.T @ (q1[iq:iq + 3] - q0[iq:iq + 3]))
            out[iv:iv + 3], out[iv + 3:iv + 6] = v, w
    return out


def qr_pivoting(tau, W_e, params_r, tol_qr=1e-8):
    """src/figaroh/tools/qrdecomposition.py:24-86 restated: column-pivoted QR, rank = index of the first pivot that is
    NOT above tol_qr (so a full-rank W_e gives rank 0: the loop never reaches its else branch), beta and phi_b rounded to
    6 decimals, expressions as in get_baseParams.  Returns (W_b, {expression: phi})."""
    from scipy import linalg
    Q, R, P = linalg.qr(W_e, pivoting=True, mode="economic")
    sorted_names = [params_r[P[i]] for i in range(P.shape[0])]
    rank = 0
    d = np.diag(R)
    for i in range(d.shape[0]):
        if abs(d[i]) > tol_qr:
            continue
        rank = i
        break
    R1, Q1, R2 = R[:rank, :rank], Q[:, :rank], R[:rank, rank:]
    beta = np.around(np.linalg.inv(R1) @ R2, 6)
    phi_b = np.round(np.linalg.inv(R1) @ (Q1.T @ tau), 6)
    names = regroup_strings(sorted_names[:rank], sorted_names[rank:], beta)
    return Q1 @ R1, dict(zip(names, phi_b))


# ---------------------------------------------------------------------------------------------------------------------
# test-data helper for the total-least-squares payload regressors (regressor.py:296-500, pure NumPy in the reference:
# the fixture tests/golden/tls_regressors.npz is the reference's own output)
def tls_inputs(W, z, nblocks, nbase=None):
    """Unloaded / loaded data sets for the total-least-squares payload regressors, cut from one golden config: the first
    half of the samples is the unloaded run, the second half the loaded one; the measurement vectors are the golden
    tau of those rows, scaled by per-block gains, the loaded one carrying the contribution of a payload on the
    picked columns plus noise.  Shared by the fixture generator and the tests (which rebuild W on the device)."""
    N = len(z["q_big"])
    half = N // 2
    rows_u = np.concatenate([j * N + np.arange(half) for j in range(nblocks)])
    rows_l = np.concatenate([j * N + half + np.arange(half) for j in range(nblocks)])
    keep = [i for i in range(W.shape[1]) if i not in set(z["idx_e"].tolist())]
    W_b = W[:, keep][:, z["idx_base"][:nbase]]  # nbase: only the leading base columns (few samples in the fixture)
    return half, rows_u, rows_l, W_b[rows_u], W_b[rows_l], W[rows_l]


# ---------------------------------------------------------------------------------------------------------------------
# test-data helper for calculate_base_kinematics_regressor (calibration_tools.py:1469-1561): a stand-in for the
# calibration subsystem's kinematic regressor with the features the reduction has to deal with -- columns that do not
# affect the measurements (exact zeros) and columns that are linear combinations of others
def synthetic_kinematic_regressor(q, ncols, seed):
    nposes = 40 if len(q) == 0 else len(q)
    rng = np.random.default_rng(seed + 7 * nposes)
    R = rng.standard_normal((3 * nposes, ncols))
    struct = np.random.default_rng(seed)  # the structure depends on the model only, not on the poses
    zero = struct.choice(ncols, size=max(1, ncols // 6), replace=False)
    R[:, zero] = 0.0
    free = [c for c in range(ncols) if c not in set(zero.tolist())]
    for k in range(max(1, ncols // 8)):
        a, b, c = struct.choice(free, size=3, replace=False)
        R[:, c] = 0.5 * R[:, a] - 2.0 * R[:, b]
    return R
