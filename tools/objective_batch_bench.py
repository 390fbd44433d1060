#!/usr/bin/env python3
"""Batched excitation objective: B trajectories of n_per samples of one finite-difference gradient
through objective_cond_batch against B calls of objective_cond, TIAGo (cfg3 fixture) by default.  Prints the wall time of
the Python calls (host arrays in, floats out) and of the device part alone (samples already in HBM)."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from figaroh_plus_amd import _lib  # noqa: E402
from figaroh_plus_amd.tools.excitation import base_columns, objective_cond, objective_cond_batch  # noqa: E402
from figaroh_plus_amd.tools.randomdata import sample_inputs  # noqa: E402
from figaroh_plus_amd.tools.regressor import regressor_flags  # noqa: E402
from figaroh_plus_amd.tools.robot import Robot  # noqa: E402

fixture, model = (sys.argv[1], sys.argv[2]) if len(sys.argv) > 2 else ("cfg3_tiago", "tiago")
n_per = int(sys.argv[3]) if len(sys.argv) > 3 else 2000
meta = json.load(open(os.path.join(ROOT, "tests", "golden", fixture + ".json")))
z = np.load(os.path.join(ROOT, "tests", "golden", fixture + ".npz"))
robot = Robot.from_flat(model)
param, idx_e, idx_base = meta["param"], z["idx_e"], z["idx_base"]
rng = np.random.default_rng(0)
report = {"model": model, "n_per": n_per, "base_columns": int(len(idx_base)), "rows": []}
for B in (1, 8, 64):
    trajs = [sample_inputs(robot.model, n_per, rng, 1.5, 2, 5) for _ in range(B)]
    for _ in range(2):
        got = objective_cond_batch(robot, trajs, param, idx_e, idx_base, coupling=meta["coupling"])
        one = [objective_cond(robot, *t, param, idx_e, idx_base, coupling=meta["coupling"]) for t in trajs[:2]]
    t0 = time.perf_counter()
    for _ in range(3):
        got = objective_cond_batch(robot, trajs, param, idx_e, idx_base, coupling=meta["coupling"])
    t_batch = (time.perf_counter() - t0) / 3
    t0 = time.perf_counter()
    seq = [objective_cond(robot, *t, param, idx_e, idx_base, coupling=meta["coupling"]) for t in trajs]
    t_seq = time.perf_counter() - t0
    err = max(abs(g - s) / s for g, s in zip(got, seq))
    # device part alone
    mode, flags, ft_mask = regressor_flags(param, meta["coupling"])
    dm = robot.device_model()
    cols = base_columns(dm.shape(mode, flags)[1], idx_e, idx_base)
    r = len(cols)
    d = [_lib.DeviceArray.from_host(np.concatenate([t[k] for t in trajs]).reshape(-1)) for k in range(3)]
    d_idx, d_R = _lib.DeviceArray.from_host(cols), _lib.DeviceArray((B * r * r,), np.float64)
    ts = []
    for _ in range(7):
        _lib.synchronize()
        t0 = time.perf_counter()
        _lib.regressor_tsqr_batch(dm, mode, flags, ft_mask, B, n_per, d[0], d[1], d[2], d_idx, r, None, d_R)
        _lib.synchronize()
        ts.append(time.perf_counter() - t0)
    row = {"B": B, "batch_call_ms": 1e3 * t_batch, "sequential_calls_ms": 1e3 * t_seq, "device_batch_ms": 1e3 * float(np.median(ts[2:])),
           "max_rel_diff_vs_sequential": err}
    report["rows"].append(row)
    sys.stderr.write(json.dumps(row) + "\n")
print(json.dumps(report, indent=1))
