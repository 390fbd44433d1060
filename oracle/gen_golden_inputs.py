"""Synthetic (q, v, a) generator of SURVEY.md section 8(d) (shared by the golden generator, tests and tools)."""
import numpy as np


def sample_inputs(model, N, rng, qr, vr, ar):
    q = rng.uniform(-qr, qr, (N, model.nq))
    for j in model.joints[1:]:
        if j.jtype == 2:
            th = rng.uniform(-np.pi, np.pi, N)
            q[:, j.idx_q], q[:, j.idx_q + 1] = np.cos(th), np.sin(th)
        elif j.jtype == 3:
            q[:, j.idx_q:j.idx_q + 3] = rng.uniform(-1, 1, (N, 3))
            quat = rng.standard_normal((N, 4))
            q[:, j.idx_q + 3:j.idx_q + 7] = quat / np.linalg.norm(quat, axis=1, keepdims=True)
    v = rng.uniform(-vr, vr, (N, model.nv))
    a = rng.uniform(-ar, ar, (N, model.nv))
    return q, v, a
