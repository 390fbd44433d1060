"""CPU suite, part 3: host-side glue of the drop-in API (no kernels involved)."""
import io
import contextlib
import os

import numpy as np
import pytest

import oracle_np
from conftest import ROOT


def test_standard_parameter_names_and_values(golden):
    robot = golden.robot()
    with contextlib.redirect_stdout(io.StringIO()):
        std = robot.get_standard_parameters(golden.param)
    names = golden.meta["names_std"][:len(std)]
    assert list(std.keys()) == names
    ref = golden.meta["phi_ref_raw"][:len(std)]
    assert [float(x) for x in std.values()] == [float(x) for x in ref]
    assert list(std.values())[10] == ref[10]  # YAML strings stay strings (T40X_config.yaml:17-20 quirk)


def test_model_attributes_mirror_pinocchio(golden):
    robot = golden.robot()
    m = robot.model
    d = golden.meta["dims"]
    assert (m.njoints, m.nq, m.nv) == (d["njoints"], d["nq"], d["nv"])
    assert m.names == golden.meta["joint_names"]
    assert [j for j in range(len(m.inertias.tolist())) if m.inertias.tolist()[j].mass != 0] == golden.meta["id_inertias"]
    assert m.joints[m.njoints - 1].idx_v + m.joints[m.njoints - 1].nv == m.nv
    assert robot.q0.shape == (m.nq,) and robot.v0.shape == (m.nv,)
    assert m.getJointId(m.names[2]) == 2 and m.getJointId("nope") == m.njoints


def test_flat_roundtrip(golden, tmp_path):
    from figaroh_plus_amd.model import Model
    m = golden.robot().model
    p = tmp_path / "m.json"
    m.save_flat(str(p))
    f1, f2 = m.to_flat(), Model.from_flat(str(p)).to_flat()
    for k in ("parents", "jtype", "idx_q", "idx_v", "axis", "placement", "mass", "lever", "inertia"):
        assert np.array_equal(np.asarray(f1[k]), np.asarray(f2[k])), k


def test_expression_strings_match_reference_format(golden):
    from figaroh_plus_amd.tools import qrdecomposition as qrd
    g = golden
    W = oracle_np.build_regressor_basic(g.flat(), g["q_big"], g["v_big"], g["a_big"], g.param)
    if g.coupling:
        W = oracle_np.add_coupling_TX40(W, len(g["q_big"]), g["v_big"], g["a_big"])
    W_e = np.delete(W, g["idx_e"], 1)
    params_r = g.meta["params_r"]
    res = oracle_np.base_parameters(W_e, params_r)
    idx_base = res["idx_base"]
    idx_regroup = [i for i in range(len(params_r)) if i not in set(idx_base)]
    got = qrd._expressions([params_r[i] for i in idx_base], [params_r[i] for i in idx_regroup], res["beta"])
    assert got == g.meta["params_base"]
    ib, ir = qrd._select(res["diagR"], params_r, 1e-8)
    assert ib == list(g["idx_base"]) and ir == idx_regroup
    with pytest.raises(AssertionError):
        qrd._select(res["diagR"], params_r[:-1], 1e-8)


def test_base_param_from_standard_and_index(golden):
    from figaroh_plus_amd.identification.identification_tools import base_param_from_standard, index_in_base_params
    g = golden
    std = {k: float(v) for k, v in g.params_std().items()}
    vals = np.array(base_param_from_standard(std, g.meta["params_base"]), dtype=float)
    assert np.allclose(vals, g["phi_from_std"], rtol=0, atol=1e-12)
    segs = [1, 2]
    idx = index_in_base_params(g.meta["params_base"], segs)
    for k, seg in enumerate(segs):
        if k in idx:
            for ii in idx[k]:
                assert any(tok.split("*")[-1][-len(str(seg)):] == str(seg) for tok in g.meta["params_base"][ii].split(" "))


def test_param_from_yaml_keys(golden_ur10):
    from figaroh_plus_amd.identification.identification_tools import get_param_from_yaml
    import yaml
    cfg = {"robot_params": [{"q_lim_def": 1.57, "dq_lim_def": 5.0, "fv": None, "fs": None, "Ia": None, "offset": None,
                             "Iam6": None, "fvm6": None, "fsm6": None, "N": None, "ratio_essential": None}],
           "problem_params": [{"is_external_wrench": False, "is_joint_torques": True, "force_torque": ["All"],
                               "external_wrench_offsets": False, "has_friction": False, "has_joint_offset": False,
                               "has_actuator_inertia": False, "has_coupled_wrist": False}],
           "processing_params": [{"cut_off_frequency_butterworth": 100.0, "ts": 0.01}],
           "tls_params": [{"mass_load": None, "which_body_loaded": None}]}
    cfg = yaml.safe_load(yaml.safe_dump(cfg))
    param = get_param_from_yaml(golden_ur10.robot(), cfg)
    ref = golden_ur10.param
    assert set(param) == set(ref)
    assert param["nb_samples"] == ref["nb_samples"] == 100
    for k in ("is_joint_torques", "has_friction", "force_torque", "ts"):
        assert param[k] == ref[k]


def test_regressor_flags_and_errors(golden):
    from figaroh_plus_amd import _lib
    from figaroh_plus_amd.tools.regressor import regressor_flags
    mode, flags, ft = regressor_flags(golden.param, golden.coupling)
    assert mode == (0 if golden.param["is_joint_torques"] else 1)
    assert bool(flags & _lib.FLAG_FRICTION) == bool(golden.param["has_friction"])
    assert bool(flags & _lib.FLAG_TX40) == golden.coupling
    bad = dict(golden.param, is_joint_torques=False, is_external_wrench=True, force_torque=["Fx", "bogus"])
    with pytest.raises(ValueError, match="Please enter valid parameters"):
        regressor_flags(bad)
    ok = dict(bad, force_torque=["Fx", "Mz"])
    assert regressor_flags(ok)[2] == (1 | 32)
    neither = dict(golden.param, is_joint_torques=False, is_external_wrench=False)
    with pytest.raises(UnboundLocalError):
        regressor_flags(neither)


def test_pipeline_rejects_mismatched_sample_shapes(golden_ur10):
    """IdentificationPipeline.set_samples must validate shapes before anything is uploaded (a wrong v / a /
    tau would be read past its HBM allocation by K1 / K3).  No device is needed: the checks run first."""
    from figaroh_plus_amd.pipeline import IdentificationPipeline
    g = golden_ur10
    pipe = IdentificationPipeline(g.robot(), g.param, params_std=g.params_std())
    q, v, a = g["q_big"], g["v_big"], g["a_big"]
    N = len(q)
    with pytest.raises(ValueError):
        pipe.set_samples(q, v[:, :5], a)                 # v with nv - 1 columns
    with pytest.raises(ValueError):
        pipe.set_samples(q, v, a[:N - 1])                # a with a row missing
    with pytest.raises(ValueError):
        pipe.set_samples(q, v, a, tau=np.zeros(6 * N + 1))  # tau from another run


def test_free_flyer_differentiation_matches_reference():
    """calculate_first_second_order_differentiation on the human model (free-flyer root: pin.difference = SE(3) log),
    constant and variable time steps, against the output of the reference's own function (oracle/gen_golden_extra.py;
    VERDICT r01: the mirror used to raise NotImplementedError here).  Host NumPy: no device needed."""
    from figaroh_plus_amd.identification.identification_tools import (calculate_first_second_order_differentiation,
                                                                       joint_difference)
    from figaroh_plus_amd.tools.robot import Robot
    z = np.load(os.path.join(ROOT, "tests", "golden", "human_differentiation.npz"))
    model = Robot.from_flat("human").model
    param = {"is_joint_torques": False, "is_external_wrench": True, "ts": float(z["ts"])}
    q, dq, ddq = calculate_first_second_order_differentiation(model, z["q"], param)
    assert np.array_equal(q, z["q_out"])
    assert np.abs(dq - z["dq"]).max() <= 1e-12 * np.abs(z["dq"]).max()
    assert np.abs(ddq - z["ddq"]).max() <= 1e-12 * np.abs(z["ddq"]).max()
    q, dq, ddq = calculate_first_second_order_differentiation(model, z["q"], param, dt=z["dt"])
    assert np.abs(dq - z["dq_dt"]).max() <= 1e-12 * np.abs(z["dq_dt"]).max()
    assert np.abs(ddq - z["ddq_dt"]).max() <= 1e-12 * np.abs(z["ddq_dt"]).max()
    # the free-flyer block is a genuine SE(3) difference, not q1 - q0
    d = joint_difference(model, z["q"][0], z["q"][1])
    assert np.abs(d[:3] - (z["q"][1, :3] - z["q"][0, :3])).max() > 1e-6
    assert np.abs(oracle_np.joint_difference(model.to_flat(), z["q"][0], z["q"][1]) - d).max() <= 1e-14


# ------------------------------------------------------------------------------------------------ 8f-3 SIP program
def _sip_fixture(cfg):
    z = np.load(os.path.join(ROOT, "tests", "golden", "sip_qp.npz"))
    g = {k.split("/", 1)[1]: z[k] for k in z.files if k.startswith(cfg + "/")}
    C = np.zeros((len(g["qp_a"]), len(g["qp_b"])))
    C[g["qp_C_row"], np.arange(len(g["qp_b"]))] = g["qp_C_val"]
    g["qp_C"] = C
    return g


def _kkt(G, a, C, b, meq, x, lag):
    s = C.T @ x - b
    return {"stationarity": np.abs(G @ x - a - C @ lag).max(),
            "primal": max(0.0, -s[meq:].min() if len(s) > meq else 0.0, np.abs(s[:meq]).max() if meq else 0.0),
            "dual": max(0.0, -lag[meq:].min()) if len(lag) > meq else 0.0,
            "complementarity": np.abs(lag[meq:] * s[meq:]).max() if len(s) > meq else 0.0}


def test_qp_solver_satisfies_kkt_on_random_programs():
    """The host Goldfarb-Idnani solver (identification/qp.py, the reference's quadprog.solve_qp): for a strictly
    convex program the KKT conditions are necessary and sufficient, so they pin the solution; equality rows, dependent
    and never-active constraints, drops from the active set included."""
    from figaroh_plus_amd.identification.qp import solve_qp
    rng = np.random.default_rng(11)
    removed = 0
    for trial in range(40):
        n = int(rng.integers(2, 40))
        m = int(rng.integers(0, 60))
        meq = int(rng.integers(0, min(n, m) // 2 + 1)) if m else 0
        A = rng.standard_normal((n + 5, n))
        G = A.T @ A + 0.1 * np.eye(n)
        a = 3 * rng.standard_normal(n)
        C = rng.standard_normal((n, m))
        x0 = rng.standard_normal(n)
        b = C.T @ x0 - rng.uniform(0, 1, m)
        b[:meq] = C[:, :meq].T @ x0
        x, f, xu, it, lag, iact = solve_qp(G, a, C, b, meq)
        assert max(_kkt(G, a, C, b, meq, x, lag).values()) <= 1e-9, trial
        assert abs(f - (0.5 * x @ G @ x - a @ x)) <= 1e-9 * (1 + abs(f))
        assert np.abs(G @ xu - a).max() <= 1e-9 * np.abs(a).max()
        assert sorted(iact.tolist()) == sorted((np.flatnonzero(lag != 0) + 1).tolist()) or len(iact) >= (lag != 0).sum()
        removed += int(it[1])
    assert removed > 0  # the sweep exercised the partial-step / drop branch
    # quadprog's error behaviour
    with pytest.raises(ValueError, match="positive definite"):
        solve_qp(-np.eye(2), np.zeros(2))
    with pytest.raises(ValueError, match="inconsistent"):
        solve_qp(np.eye(2), np.zeros(2), np.array([[1.0, -1.0], [0.0, 0.0]]), np.array([1.0, 1.0]))  # x0 >= 1, x0 <= -1
    x = solve_qp(np.eye(3), np.array([1.0, 2.0, 3.0]))[0]  # no constraints: the unconstrained minimiser
    assert np.allclose(x, [1, 2, 3])


@pytest.mark.parametrize("cfg", ["cfg4_talos", "cfg5_human"])
def test_sip_program_against_reference_fixture(cfg):
    """The SIP program of calculate_standard_parameters (identification_tools.py:466-572): tests/golden/sip_qp.npz holds
    what the reference's own function hands to quadprog.solve_qp (recorded by oracle/gen_golden_extra.py) and the
    solution of that program from an independent bounded-least-squares solve.  Host side of the mirror: the bound rows
    G, h, quadprog_solve_qp's regularisation / sign conventions, and the solver."""
    from figaroh_plus_amd.identification.identification_tools import quadprog_solve_qp, sip_constraints
    from figaroh_plus_amd.identification.qp import solve_qp
    g = _sip_fixture(cfg)
    G, h = sip_constraints(g["phi_ref"], g["COM_max"], g["COM_min"])
    assert np.array_equal(-G.T, g["qp_C"]) and np.array_equal(-h, g["qp_b"]) and int(g["meq"]) == 0
    x, f, xu, it, lag, iact = solve_qp(g["qp_G"], g["qp_a"], g["qp_C"], g["qp_b"], 0)
    scale = np.abs(g["phi_standard"]).max()
    assert np.abs(x - g["phi_standard"]).max() <= 1e-10 * scale
    assert max(_kkt(g["qp_G"], g["qp_a"], g["qp_C"], g["qp_b"], 0, x, lag).values()) <= 1e-12
    # through the reference's wrapper: P, q with G x <= h (qp_G = 0.5 (P + P^T) + 1e-5 I, qp_a = -q)
    n = len(x)
    P = g["qp_G"] - 1e-5 * np.eye(n)
    x2 = quadprog_solve_qp(P, -g["qp_a"], G, h)
    assert np.abs(x2 - g["phi_standard"]).max() <= 1e-10 * scale
    # an equality row through the wrapper (A x = b): total mass fixed to the reference's
    A = np.zeros((1, n))
    A[0, 9::10] = 1.0
    bm = np.array([g["phi_ref"][9::10].sum()])
    x3 = quadprog_solve_qp(P, -g["qp_a"], G, h, A, bm)
    assert abs(x3[9::10].sum() - bm[0]) <= 1e-10 * bm[0] and (G @ x3 <= h + 1e-10 * scale).all()


def test_calibration_parameter_names():
    """get_geo_offset / get_joint_offset (calibration_tools.py:252-331): six placement offsets per joint, one joint
    offset per degree of freedom named after Pinocchio's joint model (axis-aligned: RZ, PX, RUBZ ...; multi-dof joints
    number their entries from the second one)."""
    from figaroh_plus_amd.calibration.calibration_tools import get_geo_offset, get_joint_offset
    from figaroh_plus_amd.tools.robot import Robot
    m = Robot.from_flat("tx40").model
    geo = get_geo_offset(list(m.names[1:]))
    assert list(geo)[:7] == ["d_px_joint_1", "d_py_joint_1", "d_pz_joint_1", "d_phix_joint_1", "d_phiy_joint_1",
                             "d_phiz_joint_1", "d_px_joint_2"] and set(geo.values()) == {0}
    assert list(get_joint_offset(m, m.names[1:])) == ["offsetRZ_joint_%d" % k for k in range(1, 7)]
    t = Robot.from_flat("talos").model
    off = list(get_joint_offset(t, t.names[1:]))
    assert off[:7] == ["offsetFreeFlyer_root_joint"] + ["offsetFreeFlyer%d_root_joint" % k for k in range(2, 7)] + \
        ["offsetRZ_leg_left_1_joint"] and len(off) == t.nv
    # the fixture's names were produced by the reference's own get_geo_offset / get_joint_offset
    import json
    ref = json.load(open(os.path.join(ROOT, "tests", "golden", "calibration_base_regressor.json")))
    assert set(ref["tx40_full"]["paramsrand_e"]) <= set(geo)
    assert set(ref["talos_offsets"]["paramsrand_e"]) <= set(off)


def test_set_missing_params_setting_matches_reference():
    """identification_tools.py:86-165 against what the reference's own function returns, prints and does to the model
    (tests/golden/missing_params.json, oracle/gen_golden_missing_params.py)."""
    import contextlib
    import io
    import json
    import os
    import types
    from figaroh_plus_amd.identification.identification_tools import set_missing_params_setting
    with open(os.path.join(os.path.dirname(__file__), "golden", "missing_params.json")) as f:
        fixtures = json.load(f)
    for name, fx in fixtures.items():
        c = fx["input"]
        model = types.SimpleNamespace(lowerPositionLimit=np.array(c["lower"]), upperPositionLimit=np.array(c["upper"]),
                                      velocityLimit=np.array(c["velocity"]), effortLimit=np.array(c["effort"]),
                                      nq=c["nq"], nv=c["nv"])
        buf = io.StringIO()
        with contextlib.redirect_stdout(buf):
            res = set_missing_params_setting(types.SimpleNamespace(model=model), dict(c["settings"]))
        assert buf.getvalue() == fx["printed"], name
        got = {k: (v.tolist() if isinstance(v, np.ndarray) else v) for k, v in res.items()}
        assert got == fx["result"], name
        after = fx["model_after"]
        assert model.lowerPositionLimit.tolist() == after["lower"] and model.upperPositionLimit.tolist() == after["upper"]
        assert model.velocityLimit.tolist() == after["velocity"] and model.effortLimit.tolist() == after["effort"]


def test_essential_parameters_from_triangles_equals_script_loop():
    """identification_tools.essential_parameters (the loop of examples/staubli_TX40/identification.py:354-399 run on the two
    small triangles of a pass) against the oracle's statement-for-statement loop on the full W_b: same parameters dropped in
    the same order, same rounded estimates.  Host only: the triangles come from LAPACK here."""
    import oracle_np
    from figaroh_plus_amd.identification.identification_tools import essential_parameters
    rng = np.random.default_rng(12)
    counts = [400, 380, 410, 395]
    m, r = sum(counts), 9
    W_b = rng.standard_normal((m, r)) * np.array([1, 1, 1, 1, 0.05, 1, 0.02, 1, 0.01])  # three poorly excited parameters
    phi_true = rng.uniform(0.5, 2.0, r)
    noise = np.repeat([0.05, 0.1, 0.2, 0.07], counts)
    tau = W_b @ phi_true + noise * rng.standard_normal(m)
    names = ["p%d" % i for i in range(r)]
    phi_b = np.around(np.linalg.lstsq(W_b, tau, rcond=None)[0], 6)
    sig2, a = [], 0
    for n in counts:
        sig2.append(np.linalg.norm(tau[a:a + n] - W_b[a:a + n] @ phi_b) ** 2 / n)
        a += n
    phi_w, std_w = oracle_np.wls_script(W_b, tau, phi_b, counts)
    ref = oracle_np.essential_script(W_b, tau, names, std_w, sig2, counts, 30.0)
    assert ref["iterations"] >= 2
    sw = np.repeat(1.0 / np.sqrt(sig2), counts)
    R_ols = np.linalg.qr(np.c_[W_b, tau], mode="r")
    R_wls = np.linalg.qr(np.c_[W_b * sw[:, None], tau * sw], mode="r")
    got = essential_parameters(R_ols, R_wls, names, std_w, 30.0, rows_total=m)
    assert got["params_essential"] == ref["params_essential"] and got["iterations"] == ref["iterations"]
    assert np.abs(got["phi_e_ols"] - ref["phi_e_ols"]).max() <= 1.5e-6
    assert np.abs(got["phi_e_wls"] - ref["phi_e_wls"]).max() <= 1.5e-6
    assert np.abs(got["std_e_wls"] - ref["std_e_wls"]).max() <= 0.011
    assert np.abs(got["std_e_ols"] - ref["std_e_ols"]).max() <= 0.011
    # a tie for the largest std%: the script's int(i) raises; so does the mirror, instead of dropping the first silently
    tied = np.array(std_w, dtype=float)
    tied[[2, 5]] = tied.max() + 50.0
    with pytest.raises(TypeError, match="length-1"):
        essential_parameters(R_ols, R_wls, names, tied, 30.0, rows_total=m)


def test_relative_percent_zero_estimate_is_inf_without_warning():
    """std% of an estimate that rounds to exactly zero: the reference's 100 sqrt(C_ii) / |phi_i| is inf (NumPy division by
    zero, identification_tools.py:226-232); the mirrors return the same inf deliberately, without the RuntimeWarning."""
    import warnings
    from figaroh_plus_amd._host import relative_percent
    sigma, phi = np.array([0.5, 0.25, 0.0, 2.0]), np.array([2.0, 0.0, 0.0, -4.0])
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        got = relative_percent(sigma, phi)
    with np.errstate(divide="ignore", invalid="ignore"):
        want = np.round(100 * sigma / np.abs(phi), 2)
    assert np.array_equal(got, want, equal_nan=True) and np.isinf(got[1]) and np.isnan(got[2])


def test_null_rule_certificate():
    """_host.null_rule_bounds / null_rule_certified (round 6): the a-posteriori certificate of the null-pivot rule.  The
    coefficient sums are those of the regrouped triangle; a classification is certified only when every pivot is further from
    tol_qr than (1 + A) x the folded tol_qr / 64 (x 2)."""
    from figaroh_plus_amd._host import null_rule_bounds, null_rule_certified
    tol = 1e-8
    R1 = np.array([[1.0, 2.0, -3.0], [0.0, 1e-3, 4e-3], [0.0, 0.0, 5.0]])
    R2 = np.c_[R1 @ np.ones(3), np.zeros(3)]  # first dependent column = 1 b0 + 1 b1 + 1 b2, second = 0
    A_base, A_dep, xnorm = null_rule_bounds(R1, R2)
    assert xnorm == pytest.approx(np.abs(np.linalg.inv(R1)).sum(axis=1).max())
    # A_base[k] = sum |R1[:k,:k]^-1 R1[:k,k]|
    want = [0.0, 2.0, np.abs(np.linalg.solve(R1[:2, :2], R1[:2, 2])).sum()]
    assert np.allclose(A_base, want, rtol=1e-12)
    assert np.allclose(A_dep, [3.0, 0.0], rtol=1e-12)
    diag = np.array([1.0, 1e-3, 5.0, 1e-12, 0.0])
    assert null_rule_certified(diag, [0, 1, 2], [3, 4], (A_base, A_dep, xnorm), tol)
    # a base pivot 2 % above the tolerance with A = 2: margin 2e-10 < 2 (1 + 2) 1.5625e-10 -> not certified
    assert not null_rule_certified(np.array([1.0, 1.02e-8, 5.0, 1e-12, 0.0]), [0, 1, 2], [3, 4], (A_base, A_dep, xnorm), tol)
    # ... 20 % above it is
    assert null_rule_certified(np.array([1.0, 1.2e-8, 5.0, 1e-12, 0.0]), [0, 1, 2], [3, 4], (A_base, A_dep, xnorm), tol)
    # a dependent pivot just below the tolerance is not
    assert not null_rule_certified(np.array([1.0, 1e-3, 5.0, 0.95e-8, 0.0]), [0, 1, 2], [3, 4], (A_base, A_dep, xnorm), tol)
    # a dependent column with genuine content of its own (above tol / 8) may have been folded entirely: not certified
    assert not null_rule_certified(np.array([1.0, 1e-3, 5.0, 2e-9, 0.0]), [0, 1, 2], [3, 4], (A_base, A_dep, xnorm), tol)
    assert null_rule_certified(np.array([1.0, 1e-3, 5.0, 1.5e-10, 1e-13]), [0, 1, 2], [3, 4], (A_base, A_dep, xnorm), tol)
    # large regrouping coefficients: the folded 1.6e-10 could surface as a pivot above the tolerance
    assert not null_rule_certified(diag, [0, 1, 2], [3, 4], (A_base, np.array([700.0, 0.0]), xnorm), tol)
    # a spurious base column shows as a tiny R1_kk with huge projection coefficients
    R1s = np.array([[1.0, 500.0], [0.0, 1.1e-7]])
    Ab, Ad, xn = null_rule_bounds(R1s, np.zeros((2, 0)))
    assert Ab[1] == pytest.approx(500.0) and not null_rule_certified(np.array([1.0, 1.1e-7]), [0, 1], [], (Ab, Ad, xn), tol)
    # the estimates themselves: |R1^-1|_inf (tol / 64) |phi|_1 against 1e-7 |phi|_inf
    assert null_rule_certified(diag, [0, 1, 2], [3, 4], (A_base, A_dep, 1.0), tol, phi=np.array([3.0, -2.0, 1.0]))
    assert not null_rule_certified(diag, [0, 1, 2], [3, 4], (A_base, A_dep, 1e4), tol, phi=np.array([3.0, -2.0, 1.0]))
    assert null_rule_bounds(np.array([[1.0, 1.0], [0.0, 0.0]]), np.zeros((2, 0))) is None  # singular: never certified
    assert not null_rule_certified(diag, [0, 1, 2], [3, 4], None, tol)


def test_bench_spawns_rank_processes_and_relays_their_failure():
    """`python bench.py --gpus 2` with no launcher (WORLD_SIZE unset) starts two rank processes itself (bench.spawn_ranks) and exits
    with a child's non-zero code when they fail -- here because this container has no HIP device and the product has no CPU path
    (the GPU box runs the same command to completion: test_bench_spawns_its_own_ranks)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")
           and not k.startswith("FIGH_")}
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1"], env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=300)
    out = p.stdout.decode()
    if "needs a HIP device" in out:  # (the CPU container; on a GPU box the ranks run, which the GPU suite checks)
        assert p.returncode != 0 and out.count("needs a HIP device") == 2, out[-2000:]
    else:
        assert p.returncode == 0 and out.count('"n_gpus": 2') == 1, out[-2000:]
