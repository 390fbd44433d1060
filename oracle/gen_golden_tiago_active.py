"""Golden vectors for the ACTIVE-JOINT flow of the TIAGo script (runs only in the build container).

``examples/tiago/identification.py`` builds the full 24-block regressor, eliminates columns on the norms of the FULL matrix
(``get_index_eliminate(self.W, ..., tol_e=0.001)``, :319-322), and then only keeps the row blocks of the eight active joints
``act_idxv`` (torso_lift + arm_1..7, :406-424): ``decimate_data`` (:142-187) decimates tau and every column of those eight
blocks with ``signal.decimate(q=10, zero_phase=True)``, stacks them and hands the stack to ``double_QR`` (:335-337).

This script imports that file BY PATH (stub modules for pinocchio / matplotlib / the figaroh package names it imports, the
latter bound to the reference's own modules loaded by gen_golden.py) and runs the reference's ``decimate_data``,
``double_QR`` and ``relative_stdev`` on the restated per-sample regressor for seeded synthetic samples.  Two fixtures in one
file, ``tests/golden/tiago_active.{npz,json}`` (data only):

* ``dec``  N = 1500 samples, the script's flow with decimation: q, v, a, tau (N x 8), idx_e, params_r, the decimated stack
           (column sums of squares + every 7th row), tau_rf, params_base, phi_b, phi_std, std%;
* ``raw``  N = 400, ``decimate=False`` variant restricted to the active blocks (W_rf = the eight blocks of W_e stacked):
           what IdentificationPipeline(row_blocks=act_idxv) computes without a filter in between.

Usage:  python oracle/gen_golden_tiago_active.py
"""
import importlib.util
import json
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import gen_golden as gg  # stubs + the reference modules loaded by path  # noqa: E402

REF, GOLD = gg.REF, gg.GOLD

ACTIVE = ["torso_lift_joint", "arm_1_joint", "arm_2_joint", "arm_3_joint", "arm_4_joint", "arm_5_joint", "arm_6_joint",
          "arm_7_joint"]  # examples/tiago/identification.py:406-415


def load_script():
    """examples/tiago/identification.py as a module: its imports are satisfied by stubs, its functions are its own."""
    mpl = types.ModuleType("matplotlib")
    plt = types.ModuleType("matplotlib.pyplot")
    mpl.pyplot = plt
    fig = types.ModuleType("figaroh")
    fig_id = types.ModuleType("figaroh.identification")
    fig_tools = types.ModuleType("figaroh.tools")
    utils = types.ModuleType("utils")
    tt = types.ModuleType("utils.tiago_tools")
    tt.load_robot = lambda *a, **k: None
    sys.modules.update({
        "matplotlib": mpl, "matplotlib.pyplot": plt, "figaroh": fig, "figaroh.identification": fig_id,
        "figaroh.identification.identification_tools": gg.ref_idt, "figaroh.tools": fig_tools,
        "figaroh.tools.regressor": gg.ref_reg, "figaroh.tools.qrdecomposition": gg.ref_qr, "utils": utils,
        "utils.tiago_tools": tt,
    })
    spec = importlib.util.spec_from_file_location("ref_tiago_script", os.path.join(REF, "examples/tiago/identification.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def case(script, robot, param, params_std, ps, N, seed, decimate):
    model = robot.model
    rng = np.random.default_rng(seed)
    q, v, a = gg.sample_inputs(model, N, rng, 1.5, 2, 5)
    W = gg.ref_reg.build_regressor_basic(robot, q, v, a, param)
    idx_e, params_r = gg.ref_reg.get_index_eliminate(W, params_std, tol_e=0.001)  # on the FULL matrix, as the script
    W_e = gg.ref_reg.build_regressor_reduced(W, idx_e)
    phi_std_vec = np.array(list(params_std.values()), dtype=float)
    act = ps["act_idxv"]
    tau_full = (W @ phi_std_vec).reshape(model.nv, N)
    tau = np.stack([tau_full[j] for j in act], axis=1)  # N x 8, one column per active joint (script: eff[cols])
    tau = tau + 0.05 * rng.standard_normal(tau.shape)
    if decimate:
        t = (0.01 * np.arange(N)).reshape(-1, 1)
        t_dec, tau_dec, tau_rf, W_rf = script.decimate_data(t, tau.copy(), W_e, ps, N)
    else:
        tau_rf = tau.T.reshape(-1)
        W_rf = np.vstack([W_e[j * N:(j + 1) * N] for j in act])
    W_b, bp_dict, params_base, phi_b, phi_std = gg.ref_qr.double_QR(tau_rf, W_rf, params_r, params_std)
    std = gg.ref_idt.relative_stdev(W_b, phi_b, tau_rf)
    idx_base = [params_r.index(p.split(" ")[0]) for p in params_base]
    return {
        "q": q, "v": v, "a": a, "tau": tau, "idx_e": np.array(idx_e), "tau_rf": tau_rf,
        "W_rf_colsq": np.einsum("ij,ij->j", W_rf, W_rf), "W_rf_rows": W_rf[::7], "phi_b": np.asarray(phi_b),
        "phi_std": np.asarray(phi_std), "std": std, "idx_base": np.array(idx_base), "rows": np.array([W_rf.shape[0]]),
    }, {"params_r": list(params_r), "params_base": list(params_base)}


def main():
    script = load_script()
    mname, urdf, ff, ori, yml, coupling, *_ = gg.CONFIGS["cfg3_tiago"]
    model = gg.build_model_from_urdf(os.path.join(REF, urdf), root_joint=ff)
    robot = gg.RefRobot(model)
    param = gg._param(robot, yml)
    params_std = robot.get_standard_parameters(param)
    ps = dict(param)
    ps["active_joints"] = ACTIVE
    ps["act_Jid"] = [model.getJointId(n) for n in ACTIVE]
    ps["act_idxq"] = [model.joints[j].idx_q for j in ps["act_Jid"]]
    ps["act_idxv"] = [model.joints[j].idx_v for j in ps["act_Jid"]]
    arrays, meta = {}, {"active_joints": ACTIVE, "act_idxv": ps["act_idxv"], "act_idxq": ps["act_idxq"], "tol_e": 0.001,
                        "source": "examples/tiago/identification.py:142-187,319-337,406-424 (decimate_data, double_QR)"}
    for tag, N, seed, dec in (("dec", 1500, 31, True), ("raw", 400, 32, False)):
        arr, m = case(script, robot, param, params_std, ps, N, seed, dec)
        arrays.update({tag + "_" + k: v for k, v in arr.items()})
        meta[tag] = dict(m, N=N, seed=seed, decimate=dec)
        print("%s: N=%d, stack %d x %d, %d base parameters, |phi_b| max %.3g" % (
            tag, N, arr["rows"][0], len(m["params_r"]), len(m["params_base"]), np.abs(arr["phi_b"]).max()))
    np.savez_compressed(os.path.join(GOLD, "tiago_active.npz"), **arrays)
    with open(os.path.join(GOLD, "tiago_active.json"), "w") as f:
        json.dump(meta, f, indent=1)
    print("fixture size: %.2f MB" % (os.path.getsize(os.path.join(GOLD, "tiago_active.npz")) / 1e6))


if __name__ == "__main__":
    main()
