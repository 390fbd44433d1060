#!/usr/bin/env python3
"""K1' of a free-flyer model in a given W layout: usage k1_layout_ab.py <talos|human> <dense|link-padded|link-compact> [N]
(dense = force-compact + link-compact where they apply, the pipeline's default)."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from figaroh_plus_amd import _lib
from figaroh_plus_amd.pipeline import IdentificationPipeline
from figaroh_plus_amd.tools.randomdata import sample_inputs
from figaroh_plus_amd.tools.robot import Robot

model, layout = sys.argv[1], sys.argv[2]
N = int(sys.argv[3]) if len(sys.argv) > 3 else 1_000_000
fixture = {"talos": "cfg4_talos", "human": "cfg5_human"}[model]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
meta = json.load(open(os.path.join(root, "tests", "golden", fixture + ".json")))
robot = Robot.from_flat(model)
rng = np.random.default_rng(3)
q, v, a = sample_inputs(robot.model, N, rng, 1.5, 2, 5)
pipe = IdentificationPipeline(robot, meta["param"], params_std=dict(zip(meta["names_std"], meta["phi_ref_raw"])), w_layout=layout)
pipe.set_samples(q, v, a)
pipe.set_tau_from_parameters(np.array([float(x) for x in meta["phi_ref_raw"]]), noise_std=0.05, seed=1)
for _ in range(2):
    pipe.run()
_lib.profile_enable(True, level=1)
_lib.profile_reset()
for _ in range(5):
    pipe.run()
cnt, ms = _lib.profile_get("regressor_tree")
_lib.profile_enable(False)
print("%s %s N=%d: K1' %.3f ms (%d launches)" % (model, layout, N, ms / max(cnt, 1), cnt))
