"""Golden-vector generator (runs ONLY in the build container, never on the GPU box).

Imports the reference's own hot-path files *by path* from /root/reference
(``src/figaroh/tools/regressor.py``, ``tools/qrdecomposition.py``,
``tools/robot.py``, ``identification/identification_tools.py``) with stub
``pinocchio`` / ``quadprog`` modules, the ``pinocchio`` stub being backed by
``oracle_np.joint_torque_regressor`` (Pinocchio itself is not installable
here -- see the header of oracle_np.py).  Everything downstream of the
per-sample regressor call is therefore produced by the reference's unmodified
NumPy code.  Outputs (data only, no reference source):

* ``tests/golden/<cfg>.npz`` + ``tests/golden/<cfg>.json``  -- inputs and expected outputs
* ``tests/golden/tx40_bp_5_expressions.json``  -- column 0 of the reference's committed
  ``examples/staubli_TX40/results/TX40_bp_5.csv`` (known-answer list)
* ``figaroh_plus_amd/models/<robot>.json``  -- flattened kinematic trees of the five
  BASELINE.json robots (derived numbers, not the URDFs)

Usage:  python oracle/gen_golden.py
"""
import csv
import importlib.util
import json
import os
import sys
import types

import numpy as np
import yaml

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REF = "/root/reference"
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

import oracle_np  # noqa: E402
from figaroh_plus_amd.model import build_model_from_urdf  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")
MODELS = os.path.join(ROOT, "figaroh_plus_amd", "models")

# ----------------------------------------------------------------------------- stubs
_flat_cache = {}


def _flat(model):
    key = id(model)
    if key not in _flat_cache:
        _flat_cache[key] = model.to_flat()
    return _flat_cache[key]


pin = types.ModuleType("pinocchio")
pin.computeJointTorqueRegressor = lambda model, data, q, v, a: oracle_np.joint_torque_regressor(_flat(model), q, v, a)
pin.rnea = lambda model, data, q, v, a: oracle_np.rnea(_flat(model), q, v, a)
rw = types.ModuleType("pinocchio.robot_wrapper")
rw.RobotWrapper = type("RobotWrapper", (), {})
vz = types.ModuleType("pinocchio.visualize")
vz.GepettoVisualizer = vz.MeshcatVisualizer = object
sys.modules.update({"pinocchio": pin, "pinocchio.robot_wrapper": rw, "pinocchio.visualize": vz,
                    "quadprog": types.ModuleType("quadprog")})


def _load(name, rel):
    spec = importlib.util.spec_from_file_location(name, os.path.join(REF, rel))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


ref_reg = _load("ref_regressor", "src/figaroh/tools/regressor.py")
ref_qr = _load("ref_qrdecomposition", "src/figaroh/tools/qrdecomposition.py")
ref_robot = _load("ref_robot", "src/figaroh/tools/robot.py")
ref_idt = _load("ref_identification_tools", "src/figaroh/identification/identification_tools.py")


class RefRobot:
    """What the reference's functions need from ``Robot``: .model, .data and the
    reference's own ``get_standard_parameters`` (borrowed unbound)."""

    def __init__(self, model):
        self.model = model
        self.data = None

    def get_standard_parameters(self, param):
        return ref_robot.Robot.get_standard_parameters(self, param)


def _param(robot, yaml_rel):
    with open(os.path.join(REF, yaml_rel)) as f:
        cfg = yaml.load(f, Loader=yaml.SafeLoader)
    return ref_idt.get_param_from_yaml(robot, cfg["identification"])


def _jsonable(x):
    if isinstance(x, dict):
        return {k: _jsonable(v) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_jsonable(v) for v in x]
    if isinstance(x, np.ndarray):
        return x.tolist()
    if isinstance(x, (np.floating, np.integer)):
        return x.item()
    return x


def sample_inputs(model, N, rng, qr, vr, ar):
    """Synthetic (q, v, a) as SURVEY.md section 8(d): joints uniform, continuous joints as
    (cos, sin), free-flyer p ~ U(-1,1)^3 and a normalised Gaussian quaternion."""
    q = rng.uniform(-qr, qr, (N, model.nq))
    for j in model.joints[1:]:
        if j.jtype == 2:
            th = rng.uniform(-np.pi, np.pi, N)
            q[:, j.idx_q], q[:, j.idx_q + 1] = np.cos(th), np.sin(th)
        elif j.jtype == 3:
            q[:, j.idx_q:j.idx_q + 3] = rng.uniform(-1, 1, (N, 3))
            quat = rng.standard_normal((N, 4))
            q[:, j.idx_q + 3:j.idx_q + 7] = quat / np.linalg.norm(quat, axis=1, keepdims=True)
    v = rng.uniform(-vr, vr, (N, model.nv))
    a = rng.uniform(-ar, ar, (N, model.nv))
    return q, v, a


CONFIGS = {
    # name: (model json, urdf, free-flyer, ori, yaml, coupling, N_small, N_big, (qr, vr, ar), cfg id)
    "cfg1_tx40": ("tx40", "models/staubli_tx40_description/urdf/tx40_mdh_modified.urdf", False, None,
                  "examples/staubli_TX40/config/T40X_config.yaml", True, 32, 400, (6, 10, 30), 1),
    "cfg2_ur10": ("ur10", "examples/ur10/data/robot.urdf", False, None,
                  "examples/ur10/config/ur10_config.yaml", False, 32, 400, (6, 6, 6), 2),
    "cfg3_tiago": ("tiago", "examples/tiago/urdf/tiago_48_schunk.urdf", False, None,
                   "examples/tiago/config/tiago_config.yaml", False, 4, 48, (1.5, 2, 5), 3),
    "cfg4_talos": ("talos", "examples/talos/data/talos_full_v2.urdf", True, None,
                   "examples/human/config/human_config.yaml", False, 6, 128, (1.5, 2, 5), 4),
    "cfg5_human": ("human", "models/human_description/urdf/human.urdf", True,
                   [[1, 0, 0], [0, 0, -1], [0, 1, 0]],
                   "examples/human/config/human_config.yaml", False, 6, 96, (1.5, 2, 5), 5),
}


def build(name):
    mname, urdf, ff, ori, yml, coupling, n_small, n_big, ranges, cid = CONFIGS[name]
    model = build_model_from_urdf(os.path.join(REF, urdf), root_joint=ff)
    if ori is not None:
        model.jointPlacements[model.getJointId("root_joint")].rotation = np.array(ori, dtype=float)
    os.makedirs(MODELS, exist_ok=True)
    model.save_flat(os.path.join(MODELS, mname + ".json"))
    flat = model.to_flat()
    robot = RefRobot(model)
    param = _param(robot, yml)
    params_std = robot.get_standard_parameters(param)
    if coupling and param["has_coupled_wrist"]:
        params_std["Iam6"], params_std["fvm6"], params_std["fsm6"] = param["Iam6"], param["fvm6"], param["fsm6"]
    names = list(params_std.keys())
    phi_ref = np.array([float(x) for x in params_std.values()])

    rng = np.random.default_rng(20250410 + cid)
    out = {}

    def regress(q, v, a):
        W = ref_reg.build_regressor_basic(robot, q, v, a, param)
        if coupling:
            W = ref_reg.add_coupling_TX40(W, model, None, len(q), model.nq, model.nv, model.njoints, q, v, a)
        return W

    # -- small case: the full W tensor, entry-wise comparable
    qs, vs, as_ = sample_inputs(model, n_small, rng, *ranges)
    # exercise sign(0) = 0 and an exact-zero velocity column entry
    vs[0, :] = 0.0
    Ws = regress(qs, vs, as_)
    out.update(q_small=qs, v_small=vs, a_small=as_, W_small=Ws)
    # per-sample pinocchio-layout regressor + physics cross-check for sample 1
    Y1 = oracle_np.joint_torque_regressor(flat, qs[1], vs[1], as_[1])
    out.update(Y_sample1=Y1, rnea_sample1=oracle_np.rnea(flat, qs[1], vs[1], as_[1]))

    # -- structural case: elimination + base parameters through the reference's own functions
    qb, vb, ab = sample_inputs(model, n_big, rng, *ranges)
    Wb = regress(qb, vb, ab)
    idx_e, params_r = ref_reg.get_index_eliminate(Wb, params_std, 1e-6)
    W_e = ref_reg.build_regressor_reduced(Wb, idx_e)
    W_base, params_base, idx_base = ref_qr.get_baseParams(W_e, params_r, params_std)
    idx_base2 = ref_qr.get_baseIndex(W_e, params_r)
    assert tuple(idx_base) == tuple(idx_base2)
    R = np.linalg.qr(W_e, mode="r")
    out.update(q_big=qb, v_big=vb, a_big=ab, colsq_big=np.diag(Wb.T @ Wb).copy(),
               idx_e=np.array(idx_e, dtype=np.int64), idx_base=np.array(idx_base, dtype=np.int64),
               absdiagR=np.abs(np.diag(R)), W_checksum=np.array([Wb.sum(), np.abs(Wb).sum()]))

    # -- identification on synthetic torque: double_QR / relative_stdev / lstsq / pinv
    tau0 = Wb @ phi_ref
    noise = rng.standard_normal(len(tau0)) * 0.01 * np.sqrt(np.mean(tau0 ** 2))
    tau = tau0 + noise
    W_e2, params_r2 = ref_reg.eliminate_non_dynaffect(Wb, params_std, 1e-6)
    assert params_r2 == params_r and np.array_equal(W_e2, W_e)
    # PyYAML reads '8.05e0' as str (T40X_config.yaml:17-20); the TX40 script therefore calls double_QR
    # without params_std (identification.py:238).  Coerce so the 5-tuple branch is exercised too.
    res = ref_qr.double_QR(tau, W_e, params_r, {k: float(x) for k, x in params_std.items()})
    W_b, base_parameters, params_base_dq, phi_b, phi_std = res
    assert params_base_dq == params_base
    std_ols = ref_idt.relative_stdev(W_b, phi_b, tau)
    phi_lstsq = np.around(np.linalg.lstsq(W_b, tau, rcond=None)[0], 6)
    phi_pinv = np.linalg.pinv(W_b) @ tau
    phi_from_std = np.array(ref_idt.base_param_from_standard(
        {k: float(v) for k, v in params_std.items()}, params_base), dtype=float)
    out.update(tau=tau, phi_b=np.asarray(phi_b), phi_std=np.asarray(phi_std, dtype=float), std_ols=std_ols,
               phi_lstsq=phi_lstsq, phi_pinv=phi_pinv, phi_from_std=phi_from_std,
               cond_Wb=np.array([ref_qr.cond_num(W_b), ref_qr.cond_num(W_b, "max_over_min_sigma")]))

    # -- weighted LS: (i) the script formula with a dense SIGMA (examples/staubli_TX40/identification.py:305-346
    #    is inline script code, so it is evaluated here from its mathematical definition), and
    #    (ii) the library function, dense P and all.
    nrows_joint = len(tau) // (model.nv if param["is_joint_torques"] else 6)
    nblk = len(tau) // nrows_joint
    if len(tau) <= 6000:
        sig = np.zeros(len(tau))
        for b in range(nblk):
            sl = slice(b * nrows_joint, (b + 1) * nrows_joint)
            sig[sl] = np.linalg.norm(tau[sl] - W_b[sl] @ phi_b) ** 2 / nrows_joint
        SIGMA_inv = np.linalg.inv(np.diag(sig))
        C_X = np.linalg.inv(W_b.T @ SIGMA_inv @ W_b)
        phi_wls = np.around(C_X @ W_b.T @ SIGMA_inv @ tau, 6)
        std_wls = np.round(100 * np.sqrt(np.diag(C_X)) / np.abs(phi_wls), 2)
        out.update(phi_wls_script=phi_wls, std_wls_script=std_wls)
        if param["is_joint_torques"]:
            p2 = dict(param)
            p2["idx_tau_stop"] = [(b + 1) * nrows_joint for b in range(nblk)]
            fake = types.SimpleNamespace(model=types.SimpleNamespace(nq=nblk))
            phi_wls_lib = ref_idt.weigthed_least_squares(fake, phi_b, W_b, tau, W_b @ phi_b, p2)
            out.update(phi_wls_lib=phi_wls_lib)

    np.savez_compressed(os.path.join(GOLD, name + ".npz"), **out)
    meta = {
        "config": name,
        "model": mname,
        "coupling": bool(coupling),
        "param": _jsonable(param),
        "names_std": names,
        "phi_ref_raw": _jsonable(list(params_std.values())),
        "params_r": params_r,
        "params_base": params_base,
        "n_small": n_small,
        "n_big": n_big,
        "dims": {"njoints": model.njoints, "nq": model.nq, "nv": model.nv, "cols": int(Wb.shape[1]),
                 "kept": len(params_r), "base": len(idx_base)},
        "joint_names": list(model.names),
        "id_inertias": [j for j in range(model.njoints) if model.inertias[j].mass != 0],
    }
    with open(os.path.join(GOLD, name + ".json"), "w") as f:
        json.dump(meta, f, indent=1)
    phys = np.abs(Y1 @ oracle_np.dynamic_parameters(flat).ravel() - out["rnea_sample1"]).max()
    print("%-11s W%s kept %d base %d  |W.phi-rnea|=%.2e  max dependent |Rii|=%.2e  min independent=%.2e" % (
        name, Wb.shape, len(params_r), len(idx_base), phys,
        max([x for x in out["absdiagR"] if x <= 1e-8] + [0.0]), min(x for x in out["absdiagR"] if x > 1e-8)))
    return meta


def main():
    os.makedirs(GOLD, exist_ok=True)
    metas = {n: build(n) for n in CONFIGS}
    with open(os.path.join(REF, "examples/staubli_TX40/results/TX40_bp_5.csv")) as f:
        gold = [row[0] for row in csv.reader(f)]
    with open(os.path.join(GOLD, "tx40_bp_5_expressions.json"), "w") as f:
        json.dump({"source": "examples/staubli_TX40/results/TX40_bp_5.csv column 0", "expressions": gold}, f, indent=1)
    mine = metas["cfg1_tx40"]["params_base"]
    extra = [p for p in mine if p not in gold]
    print("TX40: %d expressions, %d in committed CSV; %d/%d CSV strings reproduced verbatim; extra: %r" % (
        len(mine), len(gold), sum(g in mine for g in gold), len(gold), extra))


if __name__ == "__main__":
    main()
