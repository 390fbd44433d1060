import os, sys
ROOT = "/root/repo"
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p)
import numpy as np
import oracle_c
import test_gpu_parity as T
from figaroh_plus_amd import _lib
from figaroh_plus_amd.pipeline import IdentificationPipeline
_lib.load()
seed = int(sys.argv[1])
rng = np.random.default_rng(1000 + seed)
n = int(rng.integers(8, 31))
parents = T._random_parents(rng, n, deep=float(rng.choice([0.4, 0.6, 0.8])))
massless = tuple(int(k) for k in rng.choice(np.arange(2, n + 1), size=n // 6, replace=False))
N = 64 * 2 + int(rng.integers(1, 64))
def inputs(m, free, N):
    q = np.zeros((N, m.nq))
    for j in m.joints[1:]:
        if j.nq == 7:
            quat = rng.standard_normal((N, 4))
            q[:, :3], q[:, 3:7] = rng.uniform(-1, 1, (N, 3)), quat / np.linalg.norm(quat, axis=1)[:, None]
        elif j.nq == 2:
            th = rng.uniform(-3, 3, N)
            q[:, j.idx_q], q[:, j.idx_q + 1] = np.cos(th), np.sin(th)
        else:
            q[:, j.idx_q] = rng.uniform(-2, 2, N)
    return q, rng.uniform(-2, 2, (N, m.nv)), rng.uniform(-3, 3, (N, m.nv))
robot = T._synthetic_tree(parents, seed=seed, massless=massless)
inputs(robot.model, False, N)  # (consumes the generator as the test does)
robot = T._synthetic_tree([0] + [p + 1 for p in parents], seed=seed, massless=tuple(k + 1 for k in massless), freeflyer=True)
m = robot.model
param = dict(is_joint_torques=False, is_external_wrench=True, has_friction=False, has_actuator_inertia=False, has_joint_offset=False, force_torque=["All"])
N = 64 * 90 + int(rng.integers(1, 64))
q, v, a = inputs(m, True, N)
mode, fl, ft = oracle_c.param_flags(param, False)
W_ref = oracle_c.OracleModel(m.to_flat()).build_regressor_basic(q, v, a, mode, fl, ft)
params_std = robot.get_standard_parameters(param)
tau = W_ref @ np.array(list(params_std.values()), dtype=float) + 1e-3 * rng.standard_normal(len(W_ref))
res = {}
for layout in ("link-padded", "dense", "link-compact"):
    for npv in (True, False):
        pipe = IdentificationPipeline(robot, param, params_std=params_std, w_layout=layout, null_pivots=npv)
        pipe.set_samples(q, v, a, tau)
        pipe.run()
        o = pipe.run()
        res[(layout, npv)] = o
        print(layout, "null_pivots", npv, "residual %.12f" % o["residual_norm"], "n_base", len(o["idx_base"]), "force_ld", getattr(pipe, "_force_ld", None))
o = res[("dense", True)]
kept = [i for i in range(W_ref.shape[1]) if i not in set(o["idx_e"])]
Wb = W_ref[:, kept][:, o["idx_base"]]
phi, r2, rk, sv = np.linalg.lstsq(Wb, tau, rcond=None)
print("LAPACK residual %.12f" % np.linalg.norm(tau - Wb @ phi), "cond(W_b) %.3e" % (sv[0] / sv[-1]), "|tau|", np.linalg.norm(tau))
R = np.linalg.qr(np.c_[Wb, tau], mode="r")
print("LAPACK QR residual %.12f" % abs(R[-1, -1]))
d = o["absdiagR"]; b = np.asarray(o["idx_base"]); dep = np.setdiff1d(np.arange(len(d)), b)
print("base pivots min %.3e max %.3e; dependent max %.3e" % (d[b].min(), d[b].max(), d[dep].max() if len(dep) else 0))
