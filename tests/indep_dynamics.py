"""Test infrastructure: an inverse dynamics that shares NOTHING with the code under test.

A small hand-written branching robot (trunk -> continuous hip -> leg; trunk -> revolute shoulder -> arm -> prismatic elbow
-> fore -> fixed -> tool), optionally on a free-flyer, is described twice from the same Python literals: as URDF text (what
the product's loader parses) and as elementary 3x3 rotations / translations evaluated here.  Generalised forces are
obtained from first principles only:

    force on a subtree      F = sum_i m_i (c_i'' - g)
    moment about the origin N = sum_i d/dt (R_i I_i R_i^T w_i + c_i x m_i c_i') - c_i x m_i g
    revolute / continuous   tau = axis_w . (N - o_w x F)          prismatic   tau = axis_w . F
    free-flyer (local)      tau = [R^T F, R^T (N - p x F)]

with c_i (centre of mass), R_i (orientation) of every body along the trajectory q(t) that has velocity v and acceleration
a at t = 0, and all time derivatives taken numerically (fourth-order central differences; the free-flyer pose is
integrated with RK4 from its LOCAL velocity v + a t).  No spatial algebra, no motion subspaces, no body regressors, no
parallel-axis bookkeeping for merged links: a convention error in the URDF loader (rpy order, axis handling, joint order,
fixed-joint merging, inertial frames) or in the regressor restatements (frames of the free-flyer rows, transforms up the
tree) shows up as an O(1) discrepancy.  Used by tests/test_independent_dynamics.py.
"""
import numpy as np

G = np.array([0.0, 0.0, -9.81])


def Rx(t):
    c, s = np.cos(t), np.sin(t)
    return np.array([[1, 0, 0], [0, c, -s], [0, s, c]])


def Ry(t):
    c, s = np.cos(t), np.sin(t)
    return np.array([[c, 0, s], [0, 1, 0], [-s, 0, c]])


def Rz(t):
    c, s = np.cos(t), np.sin(t)
    return np.array([[c, -s, 0], [s, c, 0], [0, 0, 1]])


def rpy(r, p, y):  # URDF: fixed-axis roll, pitch, yaw
    return Rz(y) @ Ry(p) @ Rx(r)


def hat(w):
    return np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]])


def rot_axis(axis, th):
    """Rotation by th about the unit vector axis, from its definition (rotate the component orthogonal to the axis)."""
    a = np.asarray(axis, dtype=float)
    a = a / np.linalg.norm(a)
    P = np.outer(a, a)
    return P + np.cos(th) * (np.eye(3) - P) + np.sin(th) * hat(a)


# ---- the robot: kinematics fixed, inertial data per instance
JOINTS = [  # name, type, parent link, child link, origin xyz, origin rpy, axis (not normalised on purpose)
    ("b_shoulder", "revolute", "trunk", "arm", (0.10, -0.20, 0.30), (0.30, -0.40, 0.50), (0.0, 0.6, 1.6)),
    ("elbow", "prismatic", "arm", "fore", (0.05, 0.25, 0.10), (-0.20, 0.10, 0.70), (2.0, 0.0, 0.0)),
    ("a_hip", "continuous", "trunk", "leg", (-0.15, 0.05, -0.20), (0.10, 0.20, -0.30), (0.36, 0.48, 0.80)),
    ("tool_fix", "fixed", "fore", "tool", (0.02, -0.03, 0.12), (0.40, 0.00, -0.60), None),
]
LINKS = ["trunk", "arm", "fore", "leg", "tool"]


def random_inertials(rng):
    """Per link: mass, inertial origin (xyz, rpy) and the inertia matrix in the inertial frame (symmetric positive definite)."""
    out = {}
    for name in LINKS:
        A = rng.standard_normal((3, 3))
        I = 0.02 * (A @ A.T + 0.5 * np.eye(3))
        out[name] = dict(mass=float(rng.uniform(0.5, 3.0)), xyz=rng.uniform(-0.1, 0.1, 3), rpy=rng.uniform(-0.6, 0.6, 3),
                         I=I)
    return out


def urdf_text(inertials):
    def f(x):
        return " ".join(repr(float(t)) for t in x)

    parts = ['<?xml version="1.0"?>', '<robot name="branching3">']
    for name in LINKS:
        d = inertials[name]
        I = d["I"]
        parts.append('<link name="%s"><inertial><origin xyz="%s" rpy="%s"/><mass value="%r"/>'
                     '<inertia ixx="%r" ixy="%r" ixz="%r" iyy="%r" iyz="%r" izz="%r"/></inertial></link>' % (
                         name, f(d["xyz"]), f(d["rpy"]), float(d["mass"]), float(I[0, 0]), float(I[0, 1]), float(I[0, 2]),
                         float(I[1, 1]), float(I[1, 2]), float(I[2, 2])))
    for name, jt, parent, child, xyz, rpy_, axis in JOINTS:  # (b_shoulder is listed before a_hip on purpose)
        ax = '<axis xyz="%s"/>' % f(axis) if axis is not None else ""
        lim = '<limit lower="-2" upper="2" velocity="3" effort="50"/>' if jt in ("revolute", "prismatic") else ""
        parts.append('<joint name="%s" type="%s"><parent link="%s"/><child link="%s"/><origin xyz="%s" rpy="%s"/>%s%s</joint>'
                     % (name, jt, parent, child, f(xyz), f(rpy_), ax, lim))
    parts.append("</robot>")
    return "\n".join(parts)


# generalised coordinates in the order the reference's conventions produce (children by ascending joint name, depth first):
# [free-flyer] a_hip (continuous: cos, sin), b_shoulder, elbow
def _frames(Rb, pb, th_hip, th_sh, d_el):
    """World pose (R, p) of every link frame and of every joint frame origin / axis."""
    J = {j[0]: j for j in JOINTS}
    T = {"trunk": (Rb, pb)}

    def child_pose(parent, joint, motion_R=None, motion_p=None):
        Rp, pp = T[parent]
        _, _, _, _, xyz, rpy_, _ = J[joint]
        Ro, po = Rp @ rpy(*rpy_), pp + Rp @ np.array(xyz)
        R = Ro if motion_R is None else Ro @ motion_R
        p = po if motion_p is None else po + Ro @ motion_p
        return (Ro, po), (R, p)

    ax = {k: np.array(J[k][6]) / np.linalg.norm(J[k][6]) for k in ("a_hip", "b_shoulder", "elbow")}
    jf = {}
    jf["a_hip"], T["leg"] = child_pose("trunk", "a_hip", motion_R=rot_axis(ax["a_hip"], th_hip))
    jf["b_shoulder"], T["arm"] = child_pose("trunk", "b_shoulder", motion_R=rot_axis(ax["b_shoulder"], th_sh))
    jf["elbow"], T["fore"] = child_pose("arm", "elbow", motion_p=ax["elbow"] * d_el)
    _, T["tool"] = child_pose("fore", "tool_fix")
    axes_w = {k: jf[k][0] @ ax[k] for k in ax}
    origins_w = {k: jf[k][1] for k in ax}
    return T, axes_w, origins_w


def _base_trajectory(q, v, a, times, freeflyer):
    """(R, p) of the trunk at the given times: identity when fixed; else RK4 of R' = R hat(w_l(t)), p' = R v_l(t)."""
    if not freeflyer:
        return {t: (np.eye(3), np.zeros(3)) for t in times}
    x, y, z, w = q[3:7]
    R0 = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                   [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                   [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])
    p0 = np.array(q[:3], dtype=float)

    def rhs(t, R, p):
        return R @ hat(v[3:6] + a[3:6] * t), R @ (v[:3] + a[:3] * t)

    def integrate(t_end):
        n = 200
        h = t_end / n
        R, p, t = R0.copy(), p0.copy(), 0.0
        for _ in range(n):
            k1R, k1p = rhs(t, R, p)
            k2R, k2p = rhs(t + h / 2, R + h / 2 * k1R, p + h / 2 * k1p)
            k3R, k3p = rhs(t + h / 2, R + h / 2 * k2R, p + h / 2 * k2p)
            k4R, k4p = rhs(t + h, R + h * k3R, p + h * k3p)
            R = R + h / 6 * (k1R + 2 * k2R + 2 * k3R + k4R)
            p = p + h / 6 * (k1p + 2 * k2p + 2 * k3p + k4p)
            t += h
        return R, p

    return {t: ((R0, p0) if t == 0.0 else integrate(t)) for t in times}


def _d1(f, k, dt):  # fourth-order first derivative at sample k of a dict keyed by integer sample index
    return (-f[k + 2] + 8 * f[k + 1] - 8 * f[k - 1] + f[k - 2]) / (12 * dt)


def _d2(f, k, dt):
    return (-f[k + 2] + 16 * f[k + 1] - 30 * f[k] + 16 * f[k - 1] - f[k - 2]) / (12 * dt * dt)


def generalised_forces(inertials, q, v, a, freeflyer, dt=2e-3):
    """tau(q, v, a) of the robot with the given inertial data, in the coordinate order described above."""
    q, v, a = (np.asarray(x, dtype=float) for x in (q, v, a))
    oq, ov = (7, 6) if freeflyer else (0, 0)
    th_hip0 = np.arctan2(q[oq + 1], q[oq])
    ks = range(-4, 5)
    base = _base_trajectory(q, v, a, [k * dt for k in ks], freeflyer)
    com, Rw, Ic = {}, {}, {}
    for k in ks:
        t = k * dt
        Rb, pb = base[t]
        jt = [q0 + vv * t + 0.5 * aa * t * t for q0, vv, aa in ((th_hip0, v[ov], a[ov]), (q[oq + 2], v[ov + 1], a[ov + 1]),
                                                                (q[oq + 3], v[ov + 2], a[ov + 2]))]
        T, axes_w, origins_w = _frames(Rb, pb, *jt)
        for name in LINKS:
            R, p = T[name]
            d = inertials[name]
            com[name, k] = p + R @ d["xyz"]
            Rc = R @ rpy(*d["rpy"])
            Rw[name, k] = R
            Ic[name, k] = Rc @ d["I"] @ Rc.T
        if k == 0:
            frames0 = (T, axes_w, origins_w)
    F, Nm = {}, {}
    for name in LINKS:
        m = inertials[name]["mass"]
        c = {k: com[name, k] for k in ks}
        Lk = {}
        for k in (-2, -1, 0, 1, 2):
            Rdot = _d1({j: Rw[name, j] for j in ks}, k, dt)
            Wm = Rdot @ Rw[name, k].T
            w = 0.5 * np.array([Wm[2, 1] - Wm[1, 2], Wm[0, 2] - Wm[2, 0], Wm[1, 0] - Wm[0, 1]])
            cdot = _d1(c, k, dt) if abs(k) <= 2 else None
            Lk[k] = Ic[name, k] @ w + np.cross(c[k], m * cdot)
        F[name] = m * (_d2(c, 0, dt) - G)
        Nm[name] = _d1(Lk, 0, dt) - np.cross(c[0], m * G)
    T0, axes_w, origins_w = frames0
    subtree = {"a_hip": ["leg"], "b_shoulder": ["arm", "fore", "tool"], "elbow": ["fore", "tool"]}
    tau = np.zeros(ov + 3)
    if freeflyer:
        Rb, pb = T0["trunk"]
        Fa, Na = sum(F[n] for n in LINKS), sum(Nm[n] for n in LINKS)
        tau[:3] = Rb.T @ Fa
        tau[3:6] = Rb.T @ (Na - np.cross(pb, Fa))
    for idx, jn in enumerate(("a_hip", "b_shoulder", "elbow")):
        Fs, Ns = sum(F[n] for n in subtree[jn]), sum(Nm[n] for n in subtree[jn])
        if jn == "elbow":
            tau[ov + idx] = axes_w[jn] @ Fs
        else:
            tau[ov + idx] = axes_w[jn] @ (Ns - np.cross(origins_w[jn], Fs))
    return tau


def sample_state(rng, freeflyer):
    th = rng.uniform(-np.pi, np.pi)
    qj = [np.cos(th), np.sin(th), rng.uniform(-1.5, 1.5), rng.uniform(-0.3, 0.3)]
    if freeflyer:
        quat = rng.standard_normal(4)
        quat /= np.linalg.norm(quat)
        q = np.concatenate([rng.uniform(-1, 1, 3), quat, qj])
        nv = 9
    else:
        q = np.array(qj)
        nv = 3
    return q, rng.uniform(-2, 2, nv), rng.uniform(-5, 5, nv)
