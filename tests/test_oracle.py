"""CPU suite, part 1: the oracle against the golden vectors produced by the reference's own code."""
import json
import os

import numpy as np
import pytest

import oracle_np
from conftest import GOLD


def test_numpy_oracle_reproduces_reference_W(golden):
    g = golden
    W = oracle_np.build_regressor_basic(g.flat(), g["q_small"], g["v_small"], g["a_small"], g.param)
    if g.coupling:
        W = oracle_np.add_coupling_TX40(W, len(g["q_small"]), g["v_small"], g["a_small"])
    assert W.shape == g["W_small"].shape
    assert np.array_equal(W, g["W_small"])  # same arithmetic, same order -> bit-exact


def test_c_oracle_matches_reference_W(golden, oracle_lib):
    g = golden
    om = oracle_lib.OracleModel(g.flat())
    mode, flags, ft = oracle_lib.param_flags(g.param, g.coupling)
    W = om.build_regressor_basic(g["q_small"], g["v_small"], g["a_small"], mode, flags, ft)
    scale = np.abs(g["W_small"]).max()
    assert np.abs(W - g["W_small"]).max() <= 1e-14 * scale * 50
    # sign(0) == 0 on the zero-velocity sample (regressor.py:57)
    zero_rows = np.where(np.all(g["v_small"] == 0, axis=1))[0]
    assert len(zero_rows) == 1


def test_regressor_times_urdf_parameters_is_rnea(golden):
    """Independent physics known-answer: Y(q,v,a) . phi_urdf == RNEA(q,v,a)."""
    g = golden
    flat = g.flat()
    phi = oracle_np.dynamic_parameters(flat).ravel()
    tau = g["Y_sample1"] @ phi
    assert np.abs(tau - g["rnea_sample1"]).max() <= 1e-10 * max(1.0, np.abs(tau).max())
    q, v, a = g["q_small"][2], g["v_small"][2], g["a_small"][2]
    Y = oracle_np.joint_torque_regressor(flat, q, v, a)
    assert np.abs(Y @ phi - oracle_np.rnea(flat, q, v, a)).max() <= 1e-10 * max(1.0, np.abs(Y @ phi).max())


def test_every_regressor_column_is_rnea_of_a_unit_parameter(golden, oracle_lib):
    """tau is linear in the inertial parameters, so column 10 (i-1) + p of Y(q, v, a) must equal the recursive Newton-Euler
    torque of the model whose only non-zero parameter is (link i, slot p): every one of the 60 .. 400 columns of all five
    models pinned on an algorithm that never forms a body regressor (the restatement's backward pass of the 6 x 10
    block and the closed-form rows of the C oracle / the HIP kernels are what is being checked)."""
    g = golden
    flat = g.flat()
    nl = int(flat["njoints"]) - 1
    om = oracle_lib.OracleModel(flat)
    for s_ in (0, 3):
        q, v, a = g["q_small"][s_], g["v_small"][s_], g["a_small"][s_]
        Y = oracle_np.joint_torque_regressor(flat, q, v, a)
        Yc = om.joint_torque_regressor(q, v, a) if hasattr(om, "joint_torque_regressor") else None
        scale = max(1.0, np.abs(Y).max())
        for i in range(nl):
            for p in range(10):
                pi = np.zeros((nl, 10))
                pi[i, p] = 1.0
                col = oracle_np.rnea_with_parameters(flat, q, v, a, pi)
                assert np.abs(Y[:, 10 * i + p] - col).max() <= 1e-11 * scale, (i, p)
                if Yc is not None:
                    assert np.abs(Yc[:, 10 * i + p] - col).max() <= 1e-11 * scale, (i, p)


def test_elimination_and_base_parameters_match_reference(golden, oracle_lib):
    g = golden
    om = oracle_lib.OracleModel(g.flat())
    mode, flags, ft = oracle_lib.param_flags(g.param, g.coupling)
    W = om.build_regressor_basic(g["q_big"], g["v_big"], g["a_big"], mode, flags, ft)
    cs = oracle_lib.colsq(W)
    assert np.abs(cs - g["colsq_big"]).max() <= 1e-12 * cs.max()
    idx_e, params_r = oracle_np.get_index_eliminate(W, g.meta["names_std"], 1e-6)
    assert idx_e == list(g["idx_e"])
    assert params_r == g.meta["params_r"]
    W_e = np.delete(W, idx_e, 1)
    res = oracle_np.base_parameters(W_e, params_r, tau=g["tau"])
    assert res["idx_base"] == list(g["idx_base"])
    assert res["params_base"] == g.meta["params_base"]
    assert np.abs(res["phi_b"] - g["phi_b"]).max() <= 2e-6 * max(1.0, np.abs(g["phi_b"]).max())
    # C Householder: same rank decision
    keep = [i for i in range(W.shape[1]) if i not in set(idx_e)]
    R = oracle_lib.householder_r(W, keep)
    d = np.abs(np.diag(R))
    assert [i for i in range(len(keep)) if d[i] > 1e-8] == list(g["idx_base"])


def test_sigma_and_wls_restatements(golden):
    g = golden
    if "phi_wls_script" not in g.z.files:
        return
    flat = g.flat()
    W = oracle_np.build_regressor_basic(flat, g["q_big"], g["v_big"], g["a_big"], g.param)
    if g.coupling:
        W = oracle_np.add_coupling_TX40(W, len(g["q_big"]), g["v_big"], g["a_big"])
    keep = [i for i in range(W.shape[1]) if i not in set(g["idx_e"].tolist())]
    W_b = W[:, keep][:, g["idx_base"]]
    std = oracle_np.relative_stdev(W_b, g["phi_b"], g["tau"])
    assert np.allclose(std, g["std_ols"], rtol=0, atol=0.011)
    nblk = g.meta["dims"]["nv"] if g.param["is_joint_torques"] else 6
    n = len(g["tau"]) // nblk
    phi, s = oracle_np.wls_script(W_b, g["tau"], g["phi_b"], [n] * nblk)
    assert np.abs(phi - g["phi_wls_script"]).max() <= 2e-6 * max(1.0, np.abs(phi).max())
    if "phi_wls_lib" in g.z.files:
        stops = [(b + 1) * n for b in range(nblk)]
        phi2 = oracle_np.weigthed_least_squares(nblk, g["phi_b"], W_b, g["tau"], W_b @ g["phi_b"], stops)
        assert np.abs(phi2 - g["phi_wls_lib"]).max() <= 2e-6 * max(1.0, np.abs(phi2).max())


def test_tx40_committed_expressions_reproduced(golden_tx40):
    """Known-answer list held by the reference: examples/staubli_TX40/results/TX40_bp_5.csv column 0.
    The current reference code emits the same 60 strings plus 'Ia6' (SURVEY.md section 4)."""
    with open(os.path.join(GOLD, "tx40_bp_5_expressions.json")) as f:
        gold = json.load(f)["expressions"]
    mine = golden_tx40.meta["params_base"]
    assert len(gold) == 60 and len(mine) == 61
    assert [p for p in mine if p != "Ia6"] == gold


def test_model_anchors(golden):
    """Structural anchors of SURVEY.md appendix A.4."""
    d = golden.meta["dims"]
    expect = {"cfg1_tx40": (7, 6, 6, 87, 76, 61), "cfg2_ur10": (7, 6, 6, 84, 49, 36),
              "cfg3_tiago": (25, 34, 24, 336, 240, 179), "cfg4_talos": (34, 39, 38, 462, 330, 234),
              "cfg5_human": (41, 46, 45, 560, 190, 164)}[golden.name]
    assert (d["njoints"], d["nq"], d["nv"], d["cols"], d["kept"], d["base"]) == expect
    if golden.name == "cfg5_human":
        assert golden.meta["id_inertias"] == [1, 4, 5, 7, 9, 12, 15, 16, 19, 21, 23, 26, 27, 30, 32, 34, 37, 38, 40]
    if golden.name == "cfg2_ur10":
        assert golden.meta["params_base"][0] == ("Izz1 + 1.0*Iyy2 + 1.0*Iyy3 + 0.375401*m3 + 1.0*Iyy4 + 0.3483*mz4"
                                                " + 0.732399*m4 + 0.732399*m5 + 0.732399*m6")


def test_tx40_real_data_known_answers(oracle_lib):
    """The reference's committed measurements -> its committed results (TX40_bp_5.csv) through the oracle.
    Pins the restated Pinocchio regressor on real trajectories: same 60 expressions, phi within the CSV's
    reproduction noise (scipy filter version drift, <= 4e-4), sigma% within 1.2 %."""
    from tx40_real_common import decimate_and_filter, load_fixture, trajectories, tx40
    z, meta = load_fixture()
    g, robot, param, params_std = tx40()
    q, dq, ddq, tau = trajectories(z, robot, param, oracle_np.low_pass_filter_data)
    sel = z["row_sel"]
    assert np.array_equal(q[sel], z["q_rows"]) and np.array_equal(dq[sel], z["dq_rows"])
    assert np.array_equal(ddq[sel], z["ddq_rows"])
    assert np.allclose([q.sum(), dq.sum(), ddq.sum(), np.abs(ddq).sum()], z["qsum"], rtol=1e-13, atol=0)
    assert not ddq[:, 5].any()  # reference quirk: range(model.nq - 1) leaves the last joint's acceleration at 0
    om = oracle_lib.OracleModel(g.flat())
    W = om.build_regressor_basic(q, dq, ddq, 0, 15)
    chk = np.array([W.sum(), np.abs(W).sum(), (W * W).sum()])
    assert np.abs(chk - z["W_checksum"]).max() <= 1e-11 * np.abs(z["W_checksum"]).max()
    W_, tau_, counts = decimate_and_filter(W, tau, param, oracle_np.decimate_joint_blocks)
    assert counts == list(z["counts"])
    assert np.abs(tau_ - z["tau_dec"]).max() <= 1e-12 * np.abs(tau_).max()
    assert np.abs(W_[::97] - z["W_dec_rows"]).max() <= 1e-10 * np.abs(W_).max()
    idx_e, params_r = oracle_np.get_index_eliminate(W_, list(params_std.keys()), 0.001)
    assert params_r == meta["params_r"]
    res = oracle_np.base_parameters(np.delete(W_, idx_e, 1), params_r, tau=tau_)
    assert res["params_base"] == meta["csv_expressions"]          # the 60 committed expressions, in order
    assert np.abs(res["phi_b"] - z["phi_b"]).max() <= 2e-6
    csvv = z["csv"]
    assert np.abs(res["phi_b"] - csvv[:, 0]).max() <= 4e-4         # committed phi_OLS
    std = oracle_np.relative_stdev(res["W_b"], res["phi_b"], tau_)
    assert (np.abs(std - csvv[:, 1]) / csvv[:, 1]).max() <= 0.012  # committed sigma%
    phi_w, std_w = oracle_np.wls_script(res["W_b"], tau_, res["phi_b"], counts)
    assert np.abs(phi_w - z["phi_wls"]).max() <= 2e-6
    assert np.abs(phi_w - csvv[:, 2]).max() <= 4e-4                # committed phi_WLS


def test_se3_log_matches_matrix_logarithm():
    """oracle_np.log6 (the restated pin.difference of a free-flyer) against an independent reference: the twist read off
    scipy.linalg.logm of the 4 x 4 homogeneous matrix.  Small, generic and near-pi rotation angles."""
    from scipy.linalg import logm
    rng = np.random.default_rng(5)
    for angle in (1e-9, 1e-5, 0.3, 1.7, 3.0, np.pi - 1e-7):
        ax = rng.standard_normal(3)
        ax /= np.linalg.norm(ax)
        K = oracle_np.skew(ax)
        R = np.eye(3) + np.sin(angle) * K + (1 - np.cos(angle)) * K @ K
        p = rng.uniform(-2, 2, 3)
        v, w = oracle_np.log6(R, p)
        M = np.eye(4)
        M[:3, :3], M[:3, 3] = R, p
        L = np.real(logm(M))
        w_ref = np.array([L[2, 1], L[0, 2], L[1, 0]])
        tol = 1e-6 if np.pi - angle < 1e-3 else 1e-9  # logm itself loses digits next to pi
        assert np.abs(w - w_ref).max() <= tol and np.abs(v - L[:3, 3]).max() <= tol


def test_numpy_oracle_reproduces_reference_qr_pivoting(golden):
    """QR_pivoting (qrdecomposition.py:24-86): the restatement against the output of the reference's own function
    (oracle/gen_golden_extra.py), including the rank-0 result for a full-rank input."""
    import json
    with open(os.path.join(GOLD, "qr_pivoting.json")) as f:
        gq = json.load(f)
    if golden.name not in gq:
        pytest.skip("no QR_pivoting fixture for this config")
    ref = gq[golden.name]
    W = oracle_np.build_regressor_basic(golden.flat(), golden["q_big"], golden["v_big"], golden["a_big"], golden.param)
    if golden.coupling:
        W = oracle_np.add_coupling_TX40(W, len(golden["q_big"]), golden["v_big"], golden["a_big"])
    keep = [i for i in range(W.shape[1]) if i not in set(golden["idx_e"].tolist())]
    W_b, bp = oracle_np.qr_pivoting(golden["tau"], W[:, keep], golden.meta["params_r"])
    assert list(bp.keys()) == ref["expressions"]
    assert np.abs(np.array(list(bp.values())) - np.array(ref["phi_b"])).max() <= 1e-6
    assert list(W_b.shape) == ref["W_b_shape"]
    W_b0, bp0 = oracle_np.qr_pivoting(golden["tau"], W[:, keep][:, golden["idx_base"]],
                                      [golden.meta["params_r"][i] for i in golden["idx_base"]])
    assert list(W_b0.shape) == ref["full_rank_result"]["W_b_shape"] and len(bp0) == 0


def test_tiago_active_joint_fixture_is_consistent_with_the_oracle():
    """tests/golden/tiago_active.* (made by the reference's decimate_data / double_QR, oracle/gen_golden_tiago_active.py): the
    ``raw`` case is reproduced by the oracle's own regressor + the oracle's restatement of the elimination and of double_QR on
    the stacked active row blocks -- the fixture and the oracle agree on what "active joints" means."""
    import json
    from conftest import GOLD, Golden
    g = Golden("cfg3_tiago")
    with open(os.path.join(GOLD, "tiago_active.json")) as f:
        meta = json.load(f)
    z = np.load(os.path.join(GOLD, "tiago_active.npz"))
    act = meta["act_idxv"]
    assert act == [12, 13, 14, 15, 16, 17, 18, 19]  # torso_lift + arm_1..7 in Pinocchio's dof numbering
    q, v, a, tau = z["raw_q"], z["raw_v"], z["raw_a"], z["raw_tau"]
    N = len(q)
    W = oracle_np.build_regressor_basic(g.flat(), q, v, a, g.param)
    norms = np.einsum("ij,ij->j", W, W)
    idx_e = np.flatnonzero(norms < meta["tol_e"]).tolist()
    assert idx_e == z["raw_idx_e"].tolist()
    kept = [c for c in range(W.shape[1]) if c not in set(idx_e)]
    Wa = np.vstack([W[b * N:(b + 1) * N][:, kept] for b in act])
    assert Wa.shape[0] == int(z["raw_rows"][0])
    colsq = np.einsum("ij,ij->j", Wa, Wa)
    assert np.abs(colsq - z["raw_W_rf_colsq"]).max() <= 1e-10 * colsq.max()
    R = np.linalg.qr(Wa, mode="r")
    assert np.flatnonzero(np.abs(np.diag(R)) > 1e-8).tolist() == z["raw_idx_base"].tolist()


def test_tiago_real_data_known_answers(oracle_lib):
    """The reference's committed TIAGo measurements -> its committed, PINOCCHIO-PRODUCED result
    (examples/tiago/data/identification/dynamic/tiago_bp_19_Oct_2024_2320.csv) through the oracle: the script's flow
    (examples/tiago/identification.py: truncate, median + Butterworth filters, gradient accelerations, full configuration, motor
    constants, regressor of all 24 dofs, elimination at 1e-3 on the full matrix, decimation of the eight ACTIVE row blocks,
    double_QR) reproduces all 44 base-parameter expressions verbatim -- regrouping coefficients such as 0.113498, 0.15315,
    0.076575 are functions of the tree's geometry as Pinocchio numbers and places it (prismatic torso 13, arm 14..20, the two
    gripper fingers 21 / 22 mirrored on arm_7) -- and the identified values and standard deviations to the file's digits.  This
    pins the restated regressor of a TREE on a Pinocchio output, as TX40_bp_5.csv does for chains."""
    from tiago_real_common import load_fixture, tiago, trajectories
    z, meta = load_fixture()
    g, robot, param, params_std = tiago()
    assert abs(meta["torso_subtree_mass"] - sum(robot.model.inertias[j].mass for j in range(13, 25))) <= 1e-12
    p, v, a, tau = trajectories(z, meta, robot)
    N = len(p)
    assert np.array_equal(p[::499], z["p_rows"]) and np.array_equal(v[::499], z["v_rows"])
    assert np.array_equal(a[::499], z["a_rows"])
    mode, flags, ft = oracle_lib.param_flags(param, False)
    W = oracle_lib.OracleModel(g.flat()).build_regressor_basic(p, v, a, mode, flags, ft)  # (the C oracle: 5 870 x 24 dofs)
    chk = np.array([W.sum(), np.abs(W).sum(), (W * W).sum()])
    assert np.abs(chk - z["W_checksum"]).max() <= 1e-11 * np.abs(z["W_checksum"]).max()
    idx_e, params_r = oracle_np.get_index_eliminate(W, list(params_std.keys()), meta["tol_e"])
    assert list(idx_e) == z["idx_e"].tolist() and params_r == meta["params_r"]
    W_e = np.delete(W, idx_e, 1)
    act = meta["act_idxv"]
    W_act = np.vstack([W_e[b * N:(b + 1) * N] for b in act])
    W_list, tau_list = oracle_np.decimate_joint_blocks(W_act, tau.T.reshape(-1), len(act), q=10, stages=1)
    W_rf, tau_rf = np.vstack(W_list), np.concatenate(tau_list)
    assert np.abs(tau_rf - z["tau_rf"]).max() <= 1e-12 * np.abs(tau_rf).max()
    assert np.abs(W_rf[::29] - z["W_rf_rows"]).max() <= 1e-10 * np.abs(W_rf).max()
    res = oracle_np.base_parameters(W_rf, params_r, tau=tau_rf)
    assert res["params_base"] == meta["csv_expressions"]          # the 44 committed expressions, in order
    csvv = z["csv"]
    assert np.abs(res["phi_b"] - csvv[:, 0]).max() <= 2e-6         # committed values (6 decimals)
    std = oracle_np.relative_stdev(res["W_b"], res["phi_b"], tau_rf)
    assert np.abs(std - csvv[:, 1] / 100).max() <= 0.011           # committed 100 x std% (2 decimals)
