#!/usr/bin/env python3
"""TIAGo (near-dependent actuator-inertia columns: pivots cross tol_qr between 4e5 and 1e6 samples): the base set with and
without the null-pivot rule over sample counts and seeds; largest dependent / smallest base pivot in both modes."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from figaroh_plus_amd.pipeline import IdentificationPipeline  # noqa: E402
from figaroh_plus_amd.tools.randomdata import sample_inputs  # noqa: E402
from figaroh_plus_amd.tools.robot import Robot  # noqa: E402

with open(os.path.join(ROOT, "tests", "golden", "cfg3_tiago.json")) as f:
    meta = json.load(f)
robot = Robot.from_flat("tiago")
params_std = dict(zip(meta["names_std"], meta["phi_ref_raw"]))
phi = np.array([float(x) for x in meta["phi_ref_raw"]])
diff = 0
for N in (200000, 400000, 600000, 800000, 900000, 1000000, 1500000):
    for seed in (1, 2, 3):
        rng = np.random.default_rng(1000 * seed + N // 1000)
        q, v, a = sample_inputs(robot.model, N, rng, 1.5, 2, 5)
        res = {}
        for on in (False, True):
            pipe = IdentificationPipeline(robot, meta["param"], params_std=params_std, w_layout="block-compact", null_pivots=on)
            pipe.set_samples(q, v, a)
            pipe.set_tau_from_parameters(phi, noise_std=0.05, seed=seed)
            out = pipe.run()
            out = pipe.run()
            d = np.asarray(out["absdiagR"])
            dep = np.setdiff1d(np.arange(len(d)), out["idx_base"])
            res[on] = (out["idx_base"], d[dep].max(), d[out["idx_base"]].min(), out["phi_b"])
            del pipe
        same = res[True][0] == res[False][0]
        dphi = np.abs(res[True][3] - res[False][3]).max() if same else float("nan")
        diff += 0 if same else 1
        print("N %7d seed %d: base %d / %d %s  dependent max %.2e / %.2e  base min %.2e / %.2e  max |dphi_b| %.1e" % (
            N, seed, len(res[False][0]), len(res[True][0]), "same" if same else "DIFFERENT", res[False][1], res[True][1], res[False][2], res[True][2], dphi), flush=True)
print("differences:", diff)
