"""Test infrastructure: first-principles inverse dynamics of ANY flattened tree (the generalisation of tests/indep_dynamics.py, which
is written for one hand-made robot): world poses by composing elementary rotations / translations down the tree along the trajectory
q(t) that has velocity v and acceleration a at t = 0, then

    force on a body          F_i = m_i (c_i'' - g)
    moment about the origin  N_i = d/dt (R_i I_i R_i^T w_i + c_i x m_i c_i') - c_i x m_i g
    revolute / continuous    tau = axis_w . (N_subtree - o_w x F_subtree)        prismatic   tau = axis_w . F_subtree
    free-flyer (local frame) tau = [R^T F_subtree, R^T (N_subtree - p x F_subtree)]

with every time derivative taken NUMERICALLY (fourth-order central differences; a free-flyer pose is integrated with RK4 from its LOCAL
velocity v + a t).  No spatial algebra, no motion subspaces, no body regressors: what it shares with the code under test is the
meaning of the model's arrays (parents, joint types, axes, placements, inertials about the centre of mass) and nothing else.  Used by
tests/test_independent_dynamics.py on random trees -- revolute / prismatic / continuous joints with random axes in random order under
a fixed base or a free-flyer root."""
import numpy as np

from indep_dynamics import _d1, _d2, hat, rot_axis


def _quat_R(x, y, z, w):
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def _freeflyer_motion(q7, v6, a6, times):
    """(R, p) of a free-flyer joint's motion at the given times: RK4 of R' = R hat(w_l(t)), p' = R v_l(t), (v_l, w_l) = v + a t."""
    R0, p0 = _quat_R(*q7[3:7]), np.array(q7[:3], dtype=float)

    def rhs(t, R):
        return R @ hat(v6[3:6] + a6[3:6] * t), R @ (v6[:3] + a6[:3] * t)

    def integrate(t_end, n=240):
        h = t_end / n
        R, p, t = R0.copy(), p0.copy(), 0.0
        for _ in range(n):
            k1R, k1p = rhs(t, R)
            k2R, k2p = rhs(t + h / 2, R + h / 2 * k1R)
            k3R, k3p = rhs(t + h / 2, R + h / 2 * k2R)
            k4R, k4p = rhs(t + h, R + h * k3R)
            R = R + h / 6 * (k1R + 2 * k2R + 2 * k3R + k4R)
            p = p + h / 6 * (k1p + 2 * k2p + 2 * k3p + k4p)
            t += h
        return R, p

    return {t: ((R0, p0) if t == 0.0 else integrate(t)) for t in times}


def generalised_forces(flat, q, v, a, dt=2e-3):
    """tau (nv,) of the flattened model ``flat`` (figaroh_plus_amd.model.Model.to_flat()) at (q, v, a), coordinates in idx_v order."""
    q, v, a = (np.asarray(x, dtype=float) for x in (q, v, a))
    n = int(flat["njoints"])
    parents, jtype = np.asarray(flat["parents"]), np.asarray(flat["jtype"])
    axis, plc = np.asarray(flat["axis"], dtype=float), np.asarray(flat["placement"], dtype=float)
    iq, iv = np.asarray(flat["idx_q"]), np.asarray(flat["idx_v"])
    mass, lever = np.asarray(flat["mass"], dtype=float), np.asarray(flat["lever"], dtype=float)
    inertia = np.asarray(flat["inertia"], dtype=float).reshape(n, 3, 3)
    g = np.asarray(flat.get("gravity", [0.0, 0.0, -9.81]), dtype=float)[:3]
    ks = list(range(-4, 5))
    times = [k * dt for k in ks]
    ff = {i: _freeflyer_motion(q[iq[i]:iq[i] + 7], v[iv[i]:iv[i] + 6], a[iv[i]:iv[i] + 6], times) for i in range(1, n) if jtype[i] == 3}
    com, Rw, Ic = {}, {}, {}
    frames0 = {}
    for k in ks:
        t = k * dt
        pose = {0: (np.eye(3), np.zeros(3))}
        for i in range(1, n):
            Rp, pp = pose[int(parents[i])]
            Ro, po = Rp @ plc[i, :9].reshape(3, 3), pp + Rp @ plc[i, 9:]
            ax = axis[i] / np.linalg.norm(axis[i]) if jtype[i] != 3 else None
            if jtype[i] == 3:
                Rj, tj = ff[i][t]
            elif jtype[i] == 1:
                Rj, tj = np.eye(3), ax * (q[iq[i]] + v[iv[i]] * t + 0.5 * a[iv[i]] * t * t)
            else:
                th0 = np.arctan2(q[iq[i] + 1], q[iq[i]]) if jtype[i] == 2 else q[iq[i]]
                Rj, tj = rot_axis(ax, th0 + v[iv[i]] * t + 0.5 * a[iv[i]] * t * t), np.zeros(3)
            pose[i] = (Ro @ Rj, po + Ro @ tj)
            if k == 0:
                frames0[i] = (Ro @ ax if ax is not None else None, po, pose[i])
        for i in range(n):
            R, p = pose[i]
            com[i, k], Rw[i, k], Ic[i, k] = p + R @ lever[i], R, R @ inertia[i] @ R.T
    F, Nm = np.zeros((n, 3)), np.zeros((n, 3))
    for i in range(1, n):
        if mass[i] == 0.0 and not inertia[i].any():
            continue
        c = {k: com[i, k] for k in ks}
        Lk = {}
        for k in (-2, -1, 0, 1, 2):
            Wm = _d1({j: Rw[i, j] for j in ks}, k, dt) @ Rw[i, k].T
            w = 0.5 * np.array([Wm[2, 1] - Wm[1, 2], Wm[0, 2] - Wm[2, 0], Wm[1, 0] - Wm[0, 1]])
            Lk[k] = Ic[i, k] @ w + np.cross(c[k], mass[i] * _d1(c, k, dt))
        F[i] = mass[i] * (_d2(c, 0, dt) - g)
        Nm[i] = _d1(Lk, 0, dt) - np.cross(c[0], mass[i] * g)
    sub = {i: [i] for i in range(1, n)}  # bodies of the subtree of joint i
    for i in range(n - 1, 0, -1):
        if parents[i] > 0:
            sub[int(parents[i])] += sub[i]
    tau = np.zeros(len(v))
    for i in range(1, n):
        Fs, Ns = F[sub[i]].sum(axis=0), Nm[sub[i]].sum(axis=0)
        ax_w, o_w, (R, p) = frames0[i]
        if jtype[i] == 3:
            tau[iv[i]:iv[i] + 3] = R.T @ Fs
            tau[iv[i] + 3:iv[i] + 6] = R.T @ (Ns - np.cross(p, Fs))
        elif jtype[i] == 1:
            tau[iv[i]] = ax_w @ Fs
        else:
            tau[iv[i]] = ax_w @ (Ns - np.cross(o_w, Fs))
    return tau
