#!/usr/bin/env python3
"""Plain triangle of the fused launch against LAPACK on the two-launch W (same samples): where do they differ?"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from figaroh_plus_amd import _lib  # noqa: E402
from figaroh_plus_amd.tools.regressor import _samples_to_device  # noqa: E402
from figaroh_plus_amd.tools.robot import Robot  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
with open(os.path.join(ROOT, "tests", "golden", "cfg2_ur10.json")) as f:
    meta = json.load(f)
g = np.load(os.path.join(ROOT, "tests", "golden", "cfg2_ur10.npz"))
robot = Robot.from_flat("ur10")
rng = np.random.default_rng(1)
q, v, a = (rng.uniform(-6, 6, (N, 6)) for _ in range(3))
tau = rng.standard_normal(6 * N)
kept = np.array([c for c in range(84) if c not in set(int(x) for x in g["idx_e"])], dtype=np.int32)
n, nc = len(kept), len(kept) + 1
_, d_q, d_v, d_a = _samples_to_device(robot.model, q, v, a)
d_W = _lib.DeviceArray((6 * N * 84,), np.float64)
d_cs = _lib.DeviceArray((84,), np.float64)
d_kept = _lib.DeviceArray.from_host(kept)
d_tau = _lib.DeviceArray.from_host(tau)
d_R = _lib.DeviceArray((nc * nc,), np.float64)
for rep in range(3):
    assert _lib.regressor_tsqr_fused(robot.device_model(), 0, N, d_q, d_v, d_a, d_W, 84, d_cs, d_kept, n, d_tau, -1.0, d_R)
    R = d_R.to_host().reshape(nc, nc)
    W = d_W.to_host().reshape(6 * N, 84)
    A = np.c_[W[:, kept], tau]
    G = A.T @ A
    E = np.abs(R.T @ R - G) / np.abs(G).max()
    Rl = np.linalg.qr(A, mode="r")
    dd = np.abs(np.abs(np.diag(R)) - np.abs(np.diag(Rl))) / np.abs(np.diag(Rl)).max()
    print("rep", rep, "max |R^T R - G| / |G|max = %.3e" % E.max(), "diag rel err max %.3e at %d" % (dd.max(), dd.argmax()))
    bad = np.argwhere(E > 1e-10)
    if len(bad):
        print("  bad entries: rows", sorted(set(bad[:, 0].tolist()))[:60])
        print("  first bad column index (compact):", bad[:, 1].min(), "padded position", bad[:, 1].min() + 64 - nc)
