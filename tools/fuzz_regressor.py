#!/usr/bin/env python3
"""Randomised run of build_regressor_basic (the drop-in, host arrays in and out) against the C oracle: random trees under a fixed base
or a free-flyer root, every flag combination (friction / actuator inertia / offset), random subsets of the wrench components
(regressor.py:96-138), ragged sample counts, zero velocities (sign(0) = 0).  usage: python tools/fuzz_regressor.py [first] [count]"""
import os
import sys
import traceback

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p)
import numpy as np  # noqa: E402
import oracle_c  # noqa: E402  (checker only)
import test_gpu_parity as T  # noqa: E402
from figaroh_plus_amd import _lib  # noqa: E402
from figaroh_plus_amd.tools.regressor import build_regressor_basic, eliminate_non_dynaffect, get_index_eliminate  # noqa: E402

_lib.load()
first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
count = int(sys.argv[2]) if len(sys.argv) > 2 else 200
bad = 0
for seed in range(first, first + count):
    rng = np.random.default_rng(9000 + seed)
    try:
        n = int(rng.integers(2, 33))
        free = bool(rng.integers(2))
        parents = T._random_parents(rng, n, deep=float(rng.choice([0.3, 0.6, 0.9])))
        massless = tuple(int(k) for k in rng.choice(np.arange(2, n + 1), size=(n - 1) // 5, replace=False)) if n > 2 else ()
        robot = T._synthetic_tree(([0] + [p + 1 for p in parents]) if free else parents, seed=seed,
                                  massless=tuple(k + 1 for k in massless) if free else massless, freeflyer=free)
        m = robot.model
        comps = ["Fx", "Fy", "Fz", "Mx", "My", "Mz"]
        ft = ["All"] if rng.random() < 0.4 else [c for c in comps if rng.random() < 0.5] or ["Mz"]
        param = dict(is_joint_torques=not free, is_external_wrench=free, has_friction=bool(rng.integers(2)),
                     has_actuator_inertia=bool(rng.integers(2)), has_joint_offset=bool(rng.integers(2)), force_torque=ft if free else None)
        N = int(rng.choice([1, 2, 63, 64, 65, 127, 130, 257, 1000]))
        q = np.zeros((N, m.nq))
        for j in m.joints[1:]:
            if j.nq == 7:
                quat = rng.standard_normal((N, 4))
                q[:, :3], q[:, 3:7] = rng.uniform(-1, 1, (N, 3)), quat / np.linalg.norm(quat, axis=1)[:, None]
            elif j.nq == 2:
                th = rng.uniform(-3, 3, N)
                q[:, j.idx_q], q[:, j.idx_q + 1] = np.cos(th), np.sin(th)
            else:
                q[:, j.idx_q] = rng.uniform(-2, 2, N)
        v, a = rng.uniform(-2, 2, (N, m.nv)), rng.uniform(-3, 3, (N, m.nv))
        v[rng.integers(N)] = 0.0
        mode, fl, ftm = oracle_c.param_flags(param, False)
        W_ref = oracle_c.OracleModel(m.to_flat()).build_regressor_basic(q, v, a, mode, fl, ftm)
        W = build_regressor_basic(robot, q, v, a, param)
        assert W.shape == W_ref.shape, (seed, W.shape, W_ref.shape)
        sc = max(1e-300, np.abs(W_ref).max())
        assert np.abs(W - W_ref).max() <= 1e-12 * sc, (seed, n, free, ft, N, np.abs(W - W_ref).max() / sc)
        assert not W[W_ref == 0].any(), (seed, "structural zeros")
        params_std = robot.get_standard_parameters(param)
        colsq = (W_ref * W_ref).sum(axis=0)
        if np.abs(colsq - 1e-6).min() > 1e-9:
            idx_e, params_r = get_index_eliminate(W, params_std, 1e-6)
            assert idx_e == [int(i) for i in range(len(colsq)) if colsq[i] < 1e-6], (seed, "idx_e")
            We, pr = eliminate_non_dynaffect(W, params_std, 1e-6)
            assert pr == params_r and np.array_equal(We, np.delete(W, idx_e, 1)), (seed, "eliminate")
    except Exception:  # noqa: BLE001
        bad += 1
        print("seed", seed, "FAILED")
        traceback.print_exc(limit=2)
print("%d random models (seeds %d .. %d), %d failures" % (count, first, first + count - 1, bad))
