#!/usr/bin/env python3
"""Where the HOST spends a step: cProfile of IdentificationPipeline.run on a bench configuration (functions by own time).
usage: host_profile.py [cfg3|cfg2|cfg4] [steps]"""
import cProfile
import json
import os
import pstats
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from figaroh_plus_amd.pipeline import IdentificationPipeline
from figaroh_plus_amd.tools.randomdata import sample_inputs
from figaroh_plus_amd.tools.robot import Robot

cfg = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
fixture, model, N = {"cfg2": ("cfg2_ur10", "ur10", 1_000_000), "cfg3": ("cfg3_tiago", "tiago", 1_000_000),
                     "cfg4": ("cfg4_talos", "talos", 1_000_000)}[cfg]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
meta = json.load(open(os.path.join(root, "tests", "golden", fixture + ".json")))
robot = Robot.from_flat(model)
rng = np.random.default_rng(3)
q, v, a = (rng.uniform(-6, 6, (N, 6)) for _ in range(3)) if cfg == "cfg2" else sample_inputs(robot.model, N, rng, 1.5, 2, 5)
pipe = IdentificationPipeline(robot, meta["param"], params_std=dict(zip(meta["names_std"], meta["phi_ref_raw"])),
                              coupling=meta["coupling"], w_layout="block-compact" if cfg == "cfg3" else "dense")
pipe.set_samples(q, v, a)
pipe.set_tau_from_parameters(np.array([float(x) for x in meta["phi_ref_raw"]]), noise_std=0.05, seed=1)
wls = cfg == "cfg3"
for _ in range(3):
    pipe.run(wls=wls)
pr = cProfile.Profile()
pr.enable()
for _ in range(steps):
    pipe.run(wls=wls)
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(28)
