"""Synthetic (q, v, a) samples for benchmarks and tests (the role of ``src/figaroh/tools/randomdata.py:20-147`` in the
reference, whose generators need Pinocchio): configurations valid for every joint type of the flattened model --
continuous joints as (cos, sin), a free-flyer as position + unit quaternion -- velocities and accelerations uniform
(SURVEY.md section 8d)."""
import numpy as np


def sample_inputs(model, N, rng, q_range, v_range, a_range):
    q = rng.uniform(-q_range, q_range, (N, model.nq))
    for j in model.joints[1:]:
        if j.jtype == 2:  # continuous: (cos, sin)
            th = rng.uniform(-np.pi, np.pi, N)
            q[:, j.idx_q], q[:, j.idx_q + 1] = np.cos(th), np.sin(th)
        elif j.jtype == 3:  # free-flyer: p in [-1, 1]^3, unit quaternion (x y z w)
            q[:, j.idx_q:j.idx_q + 3] = rng.uniform(-1, 1, (N, 3))
            quat = rng.standard_normal((N, 4))
            q[:, j.idx_q + 3:j.idx_q + 7] = quat / np.linalg.norm(quat, axis=1, keepdims=True)
    v = rng.uniform(-v_range, v_range, (N, model.nv))
    a = rng.uniform(-a_range, a_range, (N, model.nv))
    return q, v, a
