#!/usr/bin/env python3
"""Records per-shard triangles produced by the HIP path (UR10 golden samples, two contiguous sample shards) and the
one-process triangle, for the CPU test that feeds the world-size-2 exchange with device-produced factors
(tests/test_dist_cpu.py).  Run on a GPU box; writes gpurun_out/r04/hip_triangles_ur10.npz (copied to tests/golden/)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import Golden  # noqa: E402
from figaroh_plus_amd.dist import shard_range  # noqa: E402
from figaroh_plus_amd.tools.qrdecomposition import rfactor  # noqa: E402
from figaroh_plus_amd.tools.regressor import build_regressor_basic  # noqa: E402

g = Golden("cfg2_ur10")
q, v, a = g["q_big"], g["v_big"], g["a_big"]
N = len(q)
tau = g["tau"].reshape(6, N)
kept = [c for c in range(84) if c not in set(int(x) for x in g["idx_e"])]
out = {"kept": np.array(kept)}
for r in range(2):
    lo, hi = shard_range(N, r, 2)
    W = build_regressor_basic(g.robot(), q[lo:hi], v[lo:hi], a[lo:hi], g.param)
    out["R_rank%d" % r] = rfactor(W, tau=np.ascontiguousarray(tau[:, lo:hi]).reshape(-1), col_idx=kept)
W = build_regressor_basic(g.robot(), q, v, a, g.param)
out["R_one_process"] = rfactor(W, tau=g["tau"], col_idx=kept)
os.makedirs(os.path.join(ROOT, "gpurun_out", "r04"), exist_ok=True)
np.savez(os.path.join(ROOT, "gpurun_out", "r04", "hip_triangles_ur10.npz"), **out)
print("recorded", {k: v.shape for k, v in out.items()})
