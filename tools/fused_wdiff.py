#!/usr/bin/env python3
"""Where the fused launch's W differs from the two-launch W (same samples), if it does."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from figaroh_plus_amd import _lib  # noqa: E402
from figaroh_plus_amd.pipeline import IdentificationPipeline  # noqa: E402
from figaroh_plus_amd.tools.robot import Robot  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
with open(os.path.join(ROOT, "tests", "golden", "cfg2_ur10.json")) as f:
    meta = json.load(f)
robot = Robot.from_flat("ur10")
rng = np.random.default_rng(1)
q, v, a = (rng.uniform(-6, 6, (N, 6)) for _ in range(3))
params_std = dict(zip(meta["names_std"], meta["phi_ref_raw"]))
phi_ref = np.array([float(x) for x in meta["phi_ref_raw"]])
Ws = []
for fuse in (False, True):
    pipe = IdentificationPipeline(robot, meta["param"], params_std=params_std, fuse=fuse)
    pipe.set_samples(q, v, a)
    pipe.set_tau_from_parameters(phi_ref, noise_std=0.05, seed=0)
    pipe.run()
    _lib.check(_lib.load().figh_memset(pipe.W.buf.ptr, 0xff, pipe.W.rows * pipe.W.ld * 8))
    pipe.run()
    W = np.empty((pipe.W.rows, pipe.W.ld))
    _lib.check(_lib.load().figh_memcpy_d2h(W.ctypes.data, pipe.W.buf.ptr, W.nbytes))
    Ws.append(W)
    print("fuse", fuse, "fused passes", pipe.fused_passes, "nan count", int(np.isnan(W).sum()))
A, B = Ws
d = np.abs(A - B)
print("max abs diff", np.nanmax(d), "scale", np.abs(A).max(), "differing entries", int((A != B).sum()), "of", A.size)
r, c = np.nonzero(A != B)
if len(r):
    print("rows (first 20):", r[:20], "cols:", sorted(set(c.tolist()))[:40])
    print("row blocks:", sorted(set((r // N).tolist())), "sample idx mod 64 (first 20):", (r % N % 64)[:20])
    k = np.argmax(d[r, c])
    print("largest:", r[k], c[k], A[r[k], c[k]], B[r[k], c[k]])
