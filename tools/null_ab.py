#!/usr/bin/env python3
"""Same-box A/B of two library builds (FIGH_LIB_PATH): UR10 fused kernel and TIAGo step, null pivots on / off."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from figaroh_plus_amd import _lib  # noqa: E402
from figaroh_plus_amd.pipeline import IdentificationPipeline  # noqa: E402
from figaroh_plus_amd.tools.randomdata import sample_inputs  # noqa: E402
from figaroh_plus_amd.tools.robot import Robot  # noqa: E402


def run(cfg, model, N, layout, on, scopes):
    with open(os.path.join(ROOT, "tests", "golden", cfg + ".json")) as f:
        meta = json.load(f)
    robot = Robot.from_flat(model)
    params_std = dict(zip(meta["names_std"], meta["phi_ref_raw"]))
    rng = np.random.default_rng(7)
    if model == "ur10":
        q, v, a = (rng.uniform(-6, 6, (N, 6)) for _ in range(3))
    else:
        q, v, a = sample_inputs(robot.model, N, rng, 1.5, 2, 5)
    pipe = IdentificationPipeline(robot, meta["param"], params_std=params_std, w_layout=layout, null_pivots=on)
    pipe.set_samples(q, v, a)
    pipe.set_tau_from_parameters(np.array([float(x) for x in meta["phi_ref_raw"]]), noise_std=0.05, seed=0)
    for _ in range(3):
        pipe.run()
    _lib.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        pipe.run()
    _lib.synchronize()
    ms = (time.perf_counter() - t0) / 10 * 1e3
    _lib.profile_enable(True, 2)
    _lib.profile_reset()
    for _ in range(4):
        pipe.run()
    res = {}
    for s in scopes:
        cnt, tot = _lib.profile_get(s)
        if cnt:
            res[s] = round(tot / 4, 3)
    _lib.profile_enable(False)
    print("%s null=%s: step %.3f ms; per-step scope totals (ms): %s" % (cfg, on, ms, res), flush=True)
    del pipe


for on in (True, False):
    run("cfg2_ur10", "ur10", 1000000, "dense", on, ["fused_chain_tsqr", "tsqr_tree"])
for on in (True, False):
    run("cfg3_tiago", "tiago", 1000000, "block-compact", on, ["regressor_tree", "tsqr", "tsqr_group", "tsqr_reduce", "tsqr_tree",
                                                              "tsqr_block_stack"])
