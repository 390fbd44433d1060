#!/usr/bin/env python3
"""TIAGo pass at N samples with and without the null-pivot rule: the largest pivots of the dependent columns."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from figaroh_plus_amd import _lib  # noqa: E402
from figaroh_plus_amd.pipeline import IdentificationPipeline  # noqa: E402
from figaroh_plus_amd.tools.randomdata import sample_inputs  # noqa: E402
from figaroh_plus_amd.tools.robot import Robot  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
layout = sys.argv[2] if len(sys.argv) > 2 else "block-compact"
with open(os.path.join(ROOT, "tests", "golden", "cfg3_tiago.json")) as f:
    meta = json.load(f)
robot = Robot.from_flat("tiago")
params_std = dict(zip(meta["names_std"], meta["phi_ref_raw"]))
rng = np.random.default_rng(20250410 + 3)
q, v, a = sample_inputs(robot.model, N, rng, 1.5, 2, 5)
res = {}
for on in (False, True):
    pipe = IdentificationPipeline(robot, meta["param"], params_std=params_std, w_layout=layout, null_pivots=on)
    pipe.set_samples(q, v, a)
    pipe.set_tau_from_parameters(np.array([float(x) for x in meta["phi_ref_raw"]]), noise_std=0.05, seed=0)
    for _ in range(2):
        out = pipe.run()
    d = np.asarray(out["absdiagR"])
    dep = np.setdiff1d(np.arange(len(d)), out["idx_base"])
    order = dep[np.argsort(-d[dep])][:10]
    res[on] = d
    print("null pivots", on, "base", len(out["idx_base"]), "largest dependent pivots:", [(int(j), out["params_r"][j], "%.2e" % d[j]) for j in order])
    del pipe
dep = np.flatnonzero(res[False] <= 1e-8)
print("ratio on/off: max %.1f median %.1f" % ((res[True][dep] / res[False][dep]).max(), np.median(res[True][dep] / res[False][dep])))
