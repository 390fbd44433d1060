"""Golden vectors from the reference's REAL data set (runs only in the build container).

The reference commits the Staubli TX40 measurements (``examples/staubli_TX40/data/{pos_read,curr}_data.csv``,
45 000 samples at 5 kHz) and the identification results it obtained from them
(``examples/staubli_TX40/results/TX40_bp_5.csv``: expression, phi_OLS, sigma%, phi_WLS, sigma%).  This script
replays ``examples/staubli_TX40/identification.py:108-346`` with the reference's OWN functions
(``low_pass_filter_data``, ``calculate_first_second_order_differentiation``, ``build_regressor_basic``,
``add_coupling_TX40``, ``eliminate_non_dynaffect``, ``double_QR``, ``relative_stdev``; scipy's ``decimate``) on top
of the restated per-sample regressor, and stores inputs, intermediates and results as fixtures.  The committed CSV
is a known-answer file for the whole path INCLUDING the Pinocchio regressor: it is reproduced to ~1e-5 (the
residual is version drift in scipy's filters between the reference's run and this container).
"""
import csv
import json
import os
import sys

import numpy as np
import pandas as pd
from scipy import signal

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import gen_golden as gg  # stubs + the reference modules loaded by path  # noqa: E402

REF, GOLD = gg.REF, gg.GOLD


def main():
    import types
    sys.modules["pinocchio"].difference = lambda model, q0, q1: q1 - q0
    mname, urdf, ff, ori, yml, coupling, *_ = gg.CONFIGS["cfg1_tx40"]
    model = gg.build_model_from_urdf(os.path.join(REF, urdf), root_joint=ff)
    robot = gg.RefRobot(model)
    param = gg._param(robot, yml)
    params_std = robot.get_standard_parameters(param)
    params_std["Iam6"], params_std["fvm6"], params_std["fsm6"] = param["Iam6"], param["fvm6"], param["fsm6"]
    idt, reg, qrd = gg.ref_idt, gg.ref_reg, gg.ref_qr

    curr = pd.read_csv(os.path.join(REF, "examples/staubli_TX40/data/curr_data.csv")).to_numpy()
    pos = pd.read_csv(os.path.join(REF, "examples/staubli_TX40/data/pos_read_data.csv")).to_numpy()
    Nr = param["N"]
    red_q = np.diag(Nr[:6]).astype(float)
    red_q[5, 4] = Nr[5]
    q_nofilt = (np.linalg.inv(red_q) @ pos.T).T
    nbutter = 4
    nbord = 5 * nbutter
    q = np.column_stack([idt.low_pass_filter_data(q_nofilt[:, i], param, nbutter) for i in range(model.nq)])
    q[:, 1] += -np.pi / 2
    q[:, 2] += np.pi / 2
    q, dq, ddq = idt.calculate_first_second_order_differentiation(model, q, param)
    N = q.shape[0]
    W = reg.build_regressor_basic(robot, q, dq, ddq, param)
    W = reg.add_coupling_TX40(W, model, None, N, model.nq, model.nv, model.njoints, q, dq, ddq)
    red_tau = np.diag(Nr[:6]).astype(float)
    red_tau[4, 5] = Nr[5]
    tau_T = red_tau @ curr.T
    tau_T = np.delete(tau_T, np.s_[0:nbord], axis=1)
    tau_T = np.delete(tau_T, np.s_[(tau_T.shape[1] - nbord):tau_T.shape[1]], axis=1)
    tau = np.asarray(tau_T).ravel()
    nj_ = tau.shape[0] // 6
    tau_list, W_list = [], []
    for i in range(model.nv):
        t = tau[i * nj_:(i + 1) * nj_]
        for _ in range(2):
            t = signal.decimate(t, q=10, zero_phase=True)
        Wj = np.zeros((t.shape[0], W.shape[1]))
        for j in range(W.shape[1]):
            col = W[i * nj_:(i * nj_ + nj_), j]
            for _ in range(2):
                col = signal.decimate(col, q=10, zero_phase=True)
            Wj[:, j] = col
        tau_list.append(t)
        W_list.append(Wj)
    counts = []
    for i in range(len(W_list)):
        keep = np.abs(W_list[i][:, i * 14 + 11]) >= param["dq_lim_def"][i]
        W_list[i], tau_list[i] = W_list[i][keep], tau_list[i][keep]
        counts.append(int(keep.sum()))
    W_, tau_ = np.vstack(W_list), np.concatenate(tau_list)
    W_e, params_r = reg.eliminate_non_dynaffect(W_, params_std, 0.001)
    W_b, base_parameters, params_base, phi_b = qrd.double_QR(tau_, W_e, params_r)
    std_ols = idt.relative_stdev(W_b, phi_b, tau_)
    phi_ols = np.around(np.linalg.lstsq(W_b, tau_, rcond=None)[0], 6)
    # weighted LS exactly as the script writes it (dense SIGMA), identification.py:305-346
    sig = np.zeros(len(tau_))
    a = 0
    for n_i in counts:
        sig[a:a + n_i] = np.linalg.norm(tau_[a:a + n_i] - W_b[a:a + n_i] @ phi_b) ** 2 / n_i
        a += n_i
    Sinv = np.linalg.inv(np.diag(sig))
    C_X = np.linalg.inv(W_b.T @ Sinv @ W_b)
    phi_wls = np.around(C_X @ W_b.T @ Sinv @ tau_, 6)
    std_wls = np.round(100 * np.sqrt(np.diag(C_X)) / np.abs(phi_wls), 2)

    with open(os.path.join(REF, "examples/staubli_TX40/results/TX40_bp_5.csv")) as f:
        gold = [row for row in csv.reader(f)]
    csv_names = [r[0] for r in gold]
    csv_vals = np.array([[float(x) for x in r[1:5]] for r in gold])
    assert csv_names == params_base, "expression list differs from the committed CSV"
    d_ols = np.abs(phi_b - csv_vals[:, 0])
    d_wls = np.abs(phi_wls - csv_vals[:, 2])
    print("TX40 real data: N=%d, decimated rows %s, %d base parameters == CSV expressions" % (N, counts, len(params_base)))
    print("  max |phi_OLS - csv| = %.2e   max |phi_WLS - csv| = %.2e   max |std_OLS - csv|/csv = %.2e" % (
        d_ols.max(), d_wls.max(), (np.abs(std_ols - csv_vals[:, 1]) / csv_vals[:, 1]).max()))
    # the measurement files as exact integers (the CSV text has <= 11 decimals): value = int / 10^11 resp. 10^9
    from decimal import Decimal

    def exact_ints(path, exp):
        with open(path) as f:
            rows = [line.split(",") for line in f.read().split("\n")[1:] if line]
        ints = np.array([[int(Decimal(t).scaleb(exp)) for t in r] for r in rows], dtype=np.int64)
        assert np.array_equal(ints / 10.0 ** exp, np.array([[float(t) for t in r] for r in rows]))
        return ints

    curr_i = exact_ints(os.path.join(REF, "examples/staubli_TX40/data/curr_data.csv"), 11)
    pos_i = exact_ints(os.path.join(REF, "examples/staubli_TX40/data/pos_read_data.csv"), 9)
    assert np.array_equal(curr_i / 1e11, curr) and np.array_equal(pos_i / 1e9, pos)
    sel = np.arange(0, N, 997)
    np.savez_compressed(os.path.join(GOLD, "tx40_real.npz"), curr_e11=curr_i, pos_e9=pos_i,
                        q_rows=q[sel], dq_rows=dq[sel], ddq_rows=ddq[sel], row_sel=sel,
                        qsum=np.array([q.sum(), dq.sum(), ddq.sum(), np.abs(ddq).sum()]),
                        tau_dec=tau_, colsq_dec=np.einsum("ij,ij->j", W_, W_), W_dec_rows=W_[::97],
                        counts=np.array(counts), phi_b=np.asarray(phi_b), std_ols=std_ols, phi_ols=phi_ols,
                        phi_wls=phi_wls, std_wls=std_wls, csv=csv_vals,
                        W_checksum=np.array([W.sum(), np.abs(W).sum(), (W * W).sum()]))
    with open(os.path.join(GOLD, "tx40_real.json"), "w") as f:
        json.dump({"source": "examples/staubli_TX40/data/*.csv + results/TX40_bp_5.csv",
                   "params_r": params_r, "params_base": params_base, "csv_expressions": csv_names,
                   "tol_e": 0.001, "nbutter": nbutter, "decimate": [10, 10]}, f, indent=1)
    print("  fixture size: %.1f MB" % (os.path.getsize(os.path.join(GOLD, "tx40_real.npz")) / 1e6))


if __name__ == "__main__":
    main()
