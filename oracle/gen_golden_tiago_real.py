"""Known-answer replay of the TIAGo script on the reference's REAL data set (runs only in the build container).

The reference commits the TIAGo measurements (``examples/tiago/data/identification/dynamic/tiago_{position,velocity,effort}.csv``:
8 022 samples of the eight measured joints) and the result it obtained from them WITH PINOCCHIO,
``tiago_bp_19_Oct_2024_2320.csv``: 44 base-parameter expressions (e.g. ``Izz14 + 1.0*Iyy15 - 0.039*mz15 + 0.016005*m15 + ...``) with
their identified values and standard deviations.  The regrouping coefficients in those strings are functions of the tree's geometry
as Pinocchio sees it -- joint numbering (torso 13, arm 14..20, gripper fingers 21 / 22, head 23 / 24), the prismatic torso, the two
fingers hanging off arm_7 with mirrored offsets, which row blocks exist -- so the file is to the TREE kernel what
``TX40_bp_5.csv`` is to the chain kernel: a Pinocchio-produced known answer for the whole path
regressor -> elimination -> active-joint decimation -> double_QR.

This script replays ``examples/tiago/identification.py`` (imported BY PATH; its own ``load_csv_data`` semantics,
``truncate_data``, ``apply_filters``, ``estimate_acceleration``, ``build_full_configuration``, ``process_torque_data``,
``decimate_data``; the reference's ``build_regressor_basic`` / ``get_index_eliminate`` / ``double_QR`` / ``relative_stdev``) on top
of the restated per-sample regressor and stores inputs, intermediates and results as fixtures:
``tests/golden/tiago_real.npz`` / ``.json`` (the measurements as exact scaled integers -- data, not source).
The file was produced without friction / actuator-inertia / offset columns (it has no fv / fs / Ia / off terms), i.e. with
those three flags of ``config/tiago_config.yaml`` off: that is what is replayed.

Usage:  python oracle/gen_golden_tiago_real.py
"""
import csv
import json
import os
import sys

import numpy as np
import pandas as pd

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import gen_golden as gg  # noqa: E402
import gen_golden_tiago_active as ga  # noqa: E402

REF, GOLD = gg.REF, gg.GOLD
DATA = os.path.join(REF, "examples/tiago/data/identification/dynamic")
# examples/tiago/identification.py:378-404 (main): what the YAML file does not hold
REDUCTION = {"torso_lift_joint": 1, "arm_1_joint": 100, "arm_2_joint": 100, "arm_3_joint": 100, "arm_4_joint": 100,
             "arm_5_joint": 336, "arm_6_joint": 336, "arm_7_joint": 336}
KMOTOR = {"torso_lift_joint": 1, "arm_1_joint": 0.136, "arm_2_joint": 0.136, "arm_3_joint": -0.087, "arm_4_joint": -0.087,
          "arm_5_joint": -0.0613, "arm_6_joint": -0.0613, "arm_7_joint": -0.0613}


def subtree_masses(model):
    """pin.computeSubtreeMasses: data.mass[j] = mass of the subtree rooted at joint j (process_torque_data uses it for the
    gravity load of the torso)."""
    n = model.njoints
    mass = np.array([model.inertias[j].mass for j in range(n)], dtype=float)
    for j in range(n - 1, 0, -1):
        mass[model.parents[j]] += mass[j]
    return mass


def main():
    script = ga.load_script()
    mname, urdf, ff, ori, yml, coupling, *_ = gg.CONFIGS["cfg3_tiago"]
    model = gg.build_model_from_urdf(os.path.join(REF, urdf), root_joint=ff)
    robot = gg.RefRobot(model)
    robot.q0 = model.neutral()
    robot.v0 = np.zeros(model.nv)
    param = gg._param(robot, yml)
    # the committed result has no fv / fs / Ia / off terms: it was produced with these flags off
    param["has_friction"] = param["has_actuator_inertia"] = param["has_joint_offset"] = False
    ps = dict(param)
    ps["active_joints"] = list(ga.ACTIVE)
    ps["reduction_ratio"], ps["kmotor"] = REDUCTION, KMOTOR
    ps["act_Jid"] = [model.getJointId(n) for n in ga.ACTIVE]
    ps["act_idxq"] = [model.joints[j].idx_q for j in ps["act_Jid"]]
    ps["act_idxv"] = [model.joints[j].idx_v for j in ps["act_Jid"]]
    params_std = robot.get_standard_parameters(ps)

    # ---- load_csv_data (:42-60) on the committed files
    pos = pd.read_csv(os.path.join(DATA, "tiago_position.csv"))
    vel = pd.read_csv(os.path.join(DATA, "tiago_velocity.csv"))
    eff = pd.read_csv(os.path.join(DATA, "tiago_effort.csv"))
    ts = pd.read_csv(os.path.join(DATA, "tiago_position.csv"), usecols=[0]).to_numpy()
    cols = {"pos": [], "vel": [], "eff": []}
    for jn in ga.ACTIVE:
        cols["pos"].extend([c for c in pos.columns if jn in c])
        cols["vel"].extend([c for c in vel.columns if jn in c])
        cols["eff"].extend([c for c in eff.columns if jn in c])
    q_raw, dq_raw, tau_raw = pos[cols["pos"]].to_numpy(), vel[cols["vel"]].to_numpy(), eff[cols["eff"]].to_numpy()
    assert q_raw.shape == dq_raw.shape == tau_raw.shape == (8022, 8)

    # ---- process_data (:254-291)
    t_, q_, dq_, tau_ = script.truncate_data(ts, q_raw, dq_raw, tau_raw.copy(), 921, 6791)
    q_f, dq_f = script.apply_filters(t_, q_, dq_)
    ddq_f = script.estimate_acceleration(t_, dq_f)
    N = q_f.shape[0]
    p, v, a = script.build_full_configuration(robot, q_f, dq_f, ddq_f, ps, N)
    # process_torque_data (:120-139) with the stub's subtree masses
    mass = subtree_masses(model)
    tau_p = tau_.copy()
    for i, jn in enumerate(ga.ACTIVE):
        tau_p[:, i] = REDUCTION[jn] * KMOTOR[jn] * tau_p[:, i]
        if jn == "torso_lift_joint":
            tau_p[:, i] += 9.81 * mass[model.getJointId(jn)]

    # ---- calc_full_regressor / calc_baseparam (:293-337)
    W = gg.ref_reg.build_regressor_basic(robot, p, v, a, ps)
    idx_e, params_r = gg.ref_reg.get_index_eliminate(W, params_std, tol_e=0.001)
    W_e = gg.ref_reg.build_regressor_reduced(W, idx_e)
    t_dec, tau_dec, tau_rf, W_rf = script.decimate_data(t_, tau_p, W_e, ps, N)
    W_b, bp_dict, params_base, phi_b, phi_std = gg.ref_qr.double_QR(tau_rf, W_rf, params_r, params_std)
    std = gg.ref_idt.relative_stdev(W_b, phi_b, tau_rf)

    with open(os.path.join(DATA, "tiago_bp_19_Oct_2024_2320.csv")) as f:
        gold = [row for row in csv.reader(f)]
    csv_names = [r[0] for r in gold]
    csv_vals = np.array([[float(x) for x in r[1:3]] for r in gold])
    same = csv_names == list(params_base)
    print("TIAGo real data: N=%d (truncated), %d kept columns, decimated stack %d x %d, %d base parameters; CSV has %d" % (
        N, len(params_r), W_rf.shape[0], W_rf.shape[1], len(params_base), len(csv_names)))
    if not same:
        for k, (x, y) in enumerate(zip(csv_names, params_base)):
            if x != y:
                print("  first difference at %d:\n    csv : %s\n    ours: %s" % (k, x, y))
                break
    assert same, "expression list differs from the committed CSV"
    d_phi = np.abs(np.asarray(phi_b) - csv_vals[:, 0])
    print("  expressions identical; max |phi_b - csv| = %.3e (max |phi| %.3g); std%% vs csv/100: max rel %.2e" % (
        d_phi.max(), np.abs(csv_vals[:, 0]).max(), (np.abs(std - csv_vals[:, 1] / 100) / np.abs(csv_vals[:, 1] / 100)).max()))

    # the measurement files as exact scaled integers (data, not source text); t, pos, vel to 1e-18 resolution is not needed:
    # the CSV text has at most 18 significant digits -- store float64 views, which reproduce pd.read_csv bit for bit
    np.savez_compressed(os.path.join(GOLD, "tiago_real.npz"), t=ts[:, 0], q=q_raw, dq=dq_raw, tau=tau_raw,
                        idx_e=np.array(idx_e), phi_b=np.asarray(phi_b), phi_std=np.asarray(phi_std), std=std,
                        tau_rf=tau_rf, W_rf_colsq=np.einsum("ij,ij->j", W_rf, W_rf), W_rf_rows=W_rf[::29],
                        p_rows=p[::499], v_rows=v[::499], a_rows=a[::499], csv=csv_vals,
                        W_checksum=np.array([W.sum(), np.abs(W).sum(), (W * W).sum()]))
    with open(os.path.join(GOLD, "tiago_real.json"), "w") as f:
        json.dump({"source": "examples/tiago/data/identification/dynamic/*.csv + tiago_bp_19_Oct_2024_2320.csv",
                   "active_joints": ga.ACTIVE, "act_idxv": ps["act_idxv"], "act_idxq": ps["act_idxq"],
                   "reduction_ratio": REDUCTION, "kmotor": KMOTOR, "truncate": [921, 6791], "tol_e": 0.001,
                   "flags_off": ["has_friction", "has_actuator_inertia", "has_joint_offset"],
                   "params_r": list(params_r), "params_base": list(params_base), "csv_expressions": csv_names,
                   "torso_subtree_mass": float(mass[model.getJointId("torso_lift_joint")])}, f, indent=1)
    print("  fixture size: %.2f MB" % (os.path.getsize(os.path.join(GOLD, "tiago_real.npz")) / 1e6))


if __name__ == "__main__":
    main()
