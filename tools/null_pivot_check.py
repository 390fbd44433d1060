#!/usr/bin/env python3
"""Null-pivot rule (figh_tsqr_null_pivot_tol): fused UR10 launch with and without it -- time, R^T R against the Gram matrix
of the materialised W, |diag R| against LAPACK, the base set."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from figaroh_plus_amd import _lib  # noqa: E402
from figaroh_plus_amd.tools.regressor import _samples_to_device  # noqa: E402
from figaroh_plus_amd.tools.robot import Robot  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
g = np.load(os.path.join(ROOT, "tests", "golden", "cfg2_ur10.npz"))
robot = Robot.from_flat("ur10")
rng = np.random.default_rng(1)
q, v, a = (rng.uniform(-3, 3, (N, 6)) for _ in range(3))
tau = rng.standard_normal(6 * N)
kept = np.array([c for c in range(84) if c not in set(int(x) for x in g["idx_e"])], dtype=np.int32)
n, nc = len(kept), len(kept) + 1
_, d_q, d_v, d_a = _samples_to_device(robot.model, q, v, a)
d_W = _lib.DeviceArray((6 * N * 84,), np.float64)
d_cs = _lib.DeviceArray((84,), np.float64)
d_kept = _lib.DeviceArray.from_host(kept)
d_tau = _lib.DeviceArray.from_host(tau)
d_R = _lib.DeviceArray((nc * nc,), np.float64)
out = {}
for tol in (0.0, 1e-8 / 64):
    _lib.tsqr_null_pivot_tol(tol)
    for rep in range(3):
        _lib.regressor_tsqr_fused(robot.device_model(), 0, N, d_q, d_v, d_a, d_W, 84, d_cs, d_kept, n, d_tau, -1.0, d_R)
    _lib.synchronize()
    t0 = time.perf_counter()
    for rep in range(20):
        _lib.regressor_tsqr_fused(robot.device_model(), 0, N, d_q, d_v, d_a, d_W, 84, d_cs, d_kept, n, d_tau, -1.0, d_R)
    _lib.synchronize()
    ms = (time.perf_counter() - t0) / 20 * 1e3
    _lib.profile_enable(True, 1)
    _lib.profile_reset()
    for rep in range(10):
        _lib.regressor_tsqr_fused(robot.device_model(), 0, N, d_q, d_v, d_a, d_W, 84, d_cs, d_kept, n, d_tau, -1.0, d_R)
    _lib.synchronize()
    cnt, tot = _lib.profile_get("fused_chain_tsqr")
    _lib.profile_enable(False)
    print("   fused kernel %.4f ms (%d launches)" % (tot / max(cnt, 1), cnt))
    R = d_R.to_host().reshape(nc, nc)
    out[tol] = R
    dg = np.abs(np.diag(R))[:n]
    print("tol %.3e: %.3f ms per fused pass; base %d of %d; dependent |R_kk| max %.3e; smallest base pivot %.3e" % (
        tol, ms, int((dg > 1e-8).sum()), n, dg[dg <= 1e-8].max() if (dg <= 1e-8).any() else 0.0, dg[dg > 1e-8].min()))
R0, R1 = out[0.0], out[1e-8 / 64]
b0 = np.abs(np.diag(R0))[:n] > 1e-8
b1 = np.abs(np.diag(R1))[:n] > 1e-8
print("same base set:", bool((b0 == b1).all()))
G0, G1 = R0.T @ R0, R1.T @ R1
print("max |R1^T R1 - R0^T R0| / max|G| = %.3e" % (np.abs(G1 - G0).max() / np.abs(G0).max()))
# base-parameter solution from either triangle: regrouped QR of the triangle, base columns first
for name, R in (("exact", R0), ("null-pivot", R1)):
    base = np.flatnonzero(b0)
    A = R[:, list(base) + [n]]
    Rb = np.linalg.qr(A, mode="r")
    phi = np.linalg.solve(Rb[:len(base), :len(base)], Rb[:len(base), -1])
    print(name, "phi[:4]", phi[:4], "resid", abs(Rb[len(base), -1]))
    out[name] = phi
print("phi rel diff %.3e" % (np.abs(out["exact"] - out["null-pivot"]).max() / np.abs(out["exact"]).max()))
if N <= 200000:
    W = d_W.to_host().reshape(6 * N, 84)
    A = np.c_[W[:, kept], tau]
    G = A.T @ A
    for name, R in (("exact", R0), ("null-pivot", R1)):
        print(name, "max |R^T R - G| / |G|max = %.3e" % (np.abs(R.T @ R - G).max() / np.abs(G).max()))
