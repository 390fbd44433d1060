"""Shared replay of the reference's real-data TX40 identification (examples/staubli_TX40/identification.py:108-233)
up to the decimated, row-filtered (W_, tau_): used by the CPU (oracle) and GPU (HIP) known-answer tests."""
import json
import os

import numpy as np

from conftest import GOLD, Golden


def load_fixture():
    z = np.load(os.path.join(GOLD, "tx40_real.npz"))
    with open(os.path.join(GOLD, "tx40_real.json")) as f:
        meta = json.load(f)
    return z, meta


def trajectories(z, robot, param, low_pass_filter_data):
    """pos / curr measurements -> (q, dq, ddq, tau) exactly as the script prepares them.  ``low_pass_filter_data``:
    the oracle's (SciPy) or the product's (device) zero-phase filter."""
    from figaroh_plus_amd.identification.identification_tools import calculate_first_second_order_differentiation
    pos, curr = z["pos_e9"] / 1e9, z["curr_e11"] / 1e11
    Nr = param["N"]
    red_q = np.diag(Nr[:6]).astype(float)
    red_q[5, 4] = Nr[5]
    q_nofilt = (np.linalg.inv(red_q) @ pos.T).T
    nbutter = 4
    nbord = 5 * nbutter
    q = np.column_stack([low_pass_filter_data(q_nofilt[:, i], param, nbutter) for i in range(6)])
    q[:, 1] += -np.pi / 2
    q[:, 2] += np.pi / 2
    q, dq, ddq = calculate_first_second_order_differentiation(robot.model, q, param)
    red_tau = np.diag(Nr[:6]).astype(float)
    red_tau[4, 5] = Nr[5]
    tau_T = red_tau @ curr.T
    tau_T = tau_T[:, nbord:tau_T.shape[1] - nbord]
    return q, dq, ddq, np.asarray(tau_T).ravel()


def decimate_and_filter(W, tau, param, decimate_joint_blocks):
    """two decimate-by-10 stages per joint block, then drop the rows where |fv_i column| < dq_lim_def[i]"""
    nj = tau.shape[0] // 6
    W_list, tau_list = decimate_joint_blocks(W[:6 * nj], tau, 6, q=10, stages=2)
    counts = []
    for i in range(6):
        keep = np.abs(W_list[i][:, i * 14 + 11]) >= param["dq_lim_def"][i]
        W_list[i], tau_list[i] = W_list[i][keep], tau_list[i][keep]
        counts.append(int(keep.sum()))
    return np.vstack(W_list), np.concatenate(tau_list), counts


def tx40():
    g = Golden("cfg1_tx40")
    return g, g.robot(), g.param, g.params_std()
