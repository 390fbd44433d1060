"""Shared replay of the reference's real-data TIAGo identification (examples/tiago/identification.py:63-139, :254-337) up to the
regressor inputs: used by the CPU (oracle) and GPU (HIP) known-answer tests against the Pinocchio-produced
``tiago_bp_19_Oct_2024_2320.csv`` (tests/golden/tiago_real.*, oracle/gen_golden_tiago_real.py)."""
import json
import os

import numpy as np

from conftest import GOLD, Golden


def load_fixture():
    z = np.load(os.path.join(GOLD, "tiago_real.npz"))
    with open(os.path.join(GOLD, "tiago_real.json")) as f:
        meta = json.load(f)
    return z, meta


def tiago():
    """(golden, robot, param with the three flags of the committed run off, params_std of that param)."""
    g = Golden("cfg3_tiago")
    param = dict(g.param, has_friction=False, has_actuator_inertia=False, has_joint_offset=False)
    robot = g.robot()
    import contextlib
    import io
    with contextlib.redirect_stdout(io.StringIO()):  # (the reference prints a warning per short fv / fs / Ia list)
        params_std = robot.get_standard_parameters(param)
    return g, robot, param, params_std


def trajectories(z, meta, robot):
    """t, q, dq, tau measurements -> (p, v, a, tau N x 8) exactly as the script prepares them: truncate_data (:113-115),
    apply_filters (:63-90: median filter 5 + Butterworth(4, 2 / 50) filtfilt, padtype odd), estimate_acceleration (:93-100:
    gradient of dq over gradient of t), build_full_configuration (:103-110), process_torque_data (:120-139)."""
    from scipy import signal
    n_i, n_f = meta["truncate"]
    t, q, dq, tau = (z[k][n_i:n_f] for k in ("t", "q", "dq", "tau"))
    b1, b2 = signal.butter(4, 2 / (100 / 2), "low")
    padlen = 3 * (max(len(b1), len(b2)) - 1)
    qf, dqf = np.zeros(q.shape), np.zeros(dq.shape)
    for j in range(dq.shape[1]):
        qf[:, j] = signal.filtfilt(b1, b2, signal.medfilt(q[:, j], 5), padtype="odd", padlen=padlen)
        dqf[:, j] = signal.filtfilt(b1, b2, signal.medfilt(dq[:, j], 5), padtype="odd", padlen=padlen)
    ddqf = np.array([np.gradient(dqf[:, j]) / np.gradient(t) for j in range(dqf.shape[1])]).T
    N = qf.shape[0]
    model = robot.model
    p = np.tile(model.neutral(), (N, 1))
    v = np.tile(np.zeros(model.nv), (N, 1))
    a = np.tile(np.zeros(model.nv), (N, 1))
    p[:, meta["act_idxq"]] = qf
    v[:, meta["act_idxv"]] = dqf
    a[:, meta["act_idxv"]] = ddqf
    tau_p = tau.copy()
    for i, jn in enumerate(meta["active_joints"]):
        tau_p[:, i] = meta["reduction_ratio"][jn] * meta["kmotor"][jn] * tau_p[:, i]
        if jn == "torso_lift_joint":
            tau_p[:, i] += 9.81 * meta["torso_subtree_mass"]
    return p, v, a, tau_p
