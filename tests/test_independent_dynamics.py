"""The regressor restatements and the URDF loader against an inverse dynamics that shares nothing with them
(tests/indep_dynamics.py: subtree momenta differentiated numerically along the trajectory, elementary rotations only).

tau is linear in the inertial parameters, so Y(q, v, a) . pi_k == tau_k for 45 instances of one kinematic structure with
random inertial data (pi_k spanning all 40 / 30 parameters) pins EVERY column of Y -- tree branches, continuous / revolute /
prismatic joints, non-unit axes, a fixed-joint merge, inertial frames with their own rotation, the free-flyer rows in the
root joint's local frame -- on physics, with the product's own URDF loader in the loop (pi_k comes from the model it
builds, tau_k from the literals the URDF text was written from)."""
import os

import numpy as np
import pytest

import indep_dynamics as idyn
import oracle_np

K = 45


def _instances(tmp_path, freeflyer):
    from figaroh_plus_amd.model import build_model_from_urdf
    rng = np.random.default_rng(7 + int(freeflyer))
    out = []
    for k in range(K):
        ine = idyn.random_inertials(rng)
        path = os.path.join(str(tmp_path), "b3_%d.urdf" % k)
        with open(path, "w") as f:
            f.write(idyn.urdf_text(ine))
        out.append((ine, build_model_from_urdf(path, root_joint=freeflyer).to_flat()))
    return out


@pytest.mark.parametrize("freeflyer", [False, True])
def test_loader_structure_of_the_handwritten_robot(tmp_path, freeflyer):
    """Joint order (children by ascending joint name although the URDF lists them the other way round), joint types,
    index bookkeeping, the merged fixed link and the root link's destination."""
    from figaroh_plus_amd.model import JT_CONTINUOUS, JT_FREEFLYER, JT_PRISMATIC, JT_REVOLUTE
    ine, flat = _instances(tmp_path, freeflyer)[0]
    o = 1 if freeflyer else 0
    assert list(flat["names"])[1:] == (["root_joint"] if freeflyer else []) + ["a_hip", "b_shoulder", "elbow"]
    assert list(flat["jtype"])[1:] == ([JT_FREEFLYER] if freeflyer else []) + [JT_CONTINUOUS, JT_REVOLUTE, JT_PRISMATIC]
    assert list(flat["parents"])[1:] == ([0] if freeflyer else []) + [o, o, o + 2]
    assert (flat["nq"], flat["nv"]) == ((11, 9) if freeflyer else (4, 3))
    m = flat["mass"]
    assert abs(m[o] - ine["trunk"]["mass"]) < 1e-15  # universe (fixed base) or root_joint (free-flyer) carries the trunk
    assert abs(m[o + 3] - (ine["fore"]["mass"] + ine["tool"]["mass"])) < 1e-15  # fixed joint: tool merged into fore
    assert abs(m[o + 1] - ine["leg"]["mass"]) < 1e-15 and abs(m[o + 2] - ine["arm"]["mass"]) < 1e-15


@pytest.mark.parametrize("freeflyer", [False, True])
def test_every_column_against_first_principles(tmp_path, freeflyer, oracle_lib):
    inst = _instances(tmp_path, freeflyer)
    flat0 = inst[0][1]
    nl = int(flat0["njoints"]) - 1
    for _, fl in inst[1:]:  # same kinematics in every instance
        assert np.array_equal(fl["placement"], flat0["placement"]) and np.array_equal(fl["axis"], flat0["axis"])
    PI = np.array([oracle_np.dynamic_parameters(fl).ravel() for _, fl in inst])      # K x 10 nl, from the LOADER's model
    assert np.linalg.matrix_rank(PI) == 10 * nl                                       # the instances span every parameter
    om = oracle_lib.OracleModel(flat0)
    rng = np.random.default_rng(11)
    for _ in range(2):
        q, v, a = idyn.sample_state(rng, freeflyer)
        TAU = np.array([idyn.generalised_forces(ine, q, v, a, freeflyer) for ine, _ in inst])  # K x nv, first principles
        scale = np.abs(TAU).max()
        for name, Y in (("numpy", oracle_np.joint_torque_regressor(flat0, q, v, a)), ("C", om.joint_torque_regressor(q, v, a))):
            err = np.abs(PI @ Y.T - TAU).max()
            assert err <= 2e-7 * scale, (name, err, scale)
            # the columns themselves: least-squares recovery of Y from the 45 instances
            Yrec = np.linalg.lstsq(PI, TAU, rcond=None)[0].T
            assert np.abs(Yrec - Y).max() <= 1e-5 * np.abs(Y).max(), name
        # the parameter-free part: the universe / fixed-base trunk contributes nothing to the joint torques, and RNEA on
        # the loader's model agrees as well
        for (ine, fl), tau in list(zip(inst, TAU))[:3]:
            assert np.abs(oracle_np.rnea(fl, q, v, a) - tau).max() <= 2e-7 * scale


# ------------------------------------------------------------------------------------------------ random trees (round 6)
def _random_tree_flat(rng, n, freeflyer, seed, massless=()):
    import test_gpu_parity as T  # (the random-tree builders of the GPU suite: pure host code)
    parents = T._random_parents(rng, n, deep=float(rng.choice([0.4, 0.6, 0.8])))
    robot = T._synthetic_tree(([0] + [p + 1 for p in parents]) if freeflyer else parents, seed=seed,
                              massless=tuple(k + 1 for k in massless) if freeflyer else massless, freeflyer=freeflyer)
    return robot.model


def _random_state(rng, m):
    q = np.zeros(m.nq)
    for j in m.joints[1:]:
        if j.nq == 7:
            quat = rng.standard_normal(4)
            q[:3], q[3:7] = rng.uniform(-1, 1, 3), quat / np.linalg.norm(quat)
        elif j.nq == 2:
            th = rng.uniform(-3, 3)
            q[j.idx_q], q[j.idx_q + 1] = np.cos(th), np.sin(th)
        else:
            q[j.idx_q] = rng.uniform(-2, 2)
    return q, rng.uniform(-2, 2, m.nv), rng.uniform(-3, 3, m.nv)


@pytest.mark.parametrize("seed", range(10))
def test_random_trees_against_first_principles(seed, oracle_lib):
    """Random trees (4 .. 12 joints; revolute / prismatic / continuous joints with random unit axes, random placements, some
    massless links; every other one under a FREE-FLYER root) against tests/indep_tree_dynamics.py -- world poses by elementary
    rotations, momenta differentiated numerically, nothing of the spatial algebra: Y(q, v, a) . pi == tau for both regressor
    restatements and RNEA.  This is what pins the free-flyer rows (the root joint's LOCAL frame, local velocities) and the
    propagation through arbitrary joint sequences on physics beyond the one hand-written robot above."""
    import indep_tree_dynamics as itd
    rng = np.random.default_rng(300 + seed)
    n = int(rng.integers(4, 13))
    free = bool(seed & 1)
    massless = tuple(int(k) for k in rng.choice(np.arange(2, n + 1), size=n // 5, replace=False))
    m = _random_tree_flat(rng, n, free, seed, massless)
    flat = m.to_flat()
    pi = oracle_np.dynamic_parameters(flat).ravel()
    om = oracle_lib.OracleModel(flat)
    for _ in range(2):
        q, v, a = _random_state(rng, m)
        tau = itd.generalised_forces(flat, q, v, a)
        scale = np.abs(tau).max()
        for name, Y in (("numpy", oracle_np.joint_torque_regressor(flat, q, v, a)), ("C", om.joint_torque_regressor(q, v, a))):
            assert np.abs(Y @ pi - tau).max() <= 5e-7 * scale, (name, seed, np.abs(Y @ pi - tau).max(), scale)
        assert np.abs(oracle_np.rnea(flat, q, v, a) - tau).max() <= 5e-7 * scale


@pytest.mark.parametrize("freeflyer", [False, True])
def test_random_tree_every_column_against_first_principles(freeflyer, oracle_lib):
    """One random tree of six joints, its kinematics fixed and its inertial data drawn 70 times (spanning all 10 parameters of every
    body): the least-squares recovery of Y from the first-principles torques pins EVERY column of the regressor, free-flyer wrench
    rows included, as the hand-written robot's test does for its one structure."""
    import indep_tree_dynamics as itd
    rng = np.random.default_rng(77 + int(freeflyer))
    m = _random_tree_flat(rng, 5 if freeflyer else 6, freeflyer, seed=9 + int(freeflyer))
    flat0 = m.to_flat()
    n = int(flat0["njoints"])
    nl = n - 1
    K = 10 * nl + 10
    inst = []
    for _ in range(K):
        fl = dict(flat0)
        fl["mass"] = np.r_[0.0, rng.uniform(0.3, 3.0, nl)]
        fl["lever"] = np.vstack([np.zeros(3), rng.uniform(-0.2, 0.2, (nl, 3))])
        I = []
        for _k in range(nl):
            A = rng.standard_normal((3, 3))
            I.append((0.02 * (A @ A.T) + 0.005 * np.eye(3)).reshape(9))
        fl["inertia"] = np.vstack([np.zeros(9), np.array(I)])
        inst.append(fl)
    PI = np.array([oracle_np.dynamic_parameters(fl).ravel() for fl in inst])
    assert np.linalg.matrix_rank(PI) == 10 * nl
    q, v, a = _random_state(rng, m)
    TAU = np.array([itd.generalised_forces(fl, q, v, a) for fl in inst])
    scale = np.abs(TAU).max()
    om = oracle_lib.OracleModel(flat0)
    for name, Y in (("numpy", oracle_np.joint_torque_regressor(flat0, q, v, a)), ("C", om.joint_torque_regressor(q, v, a))):
        assert np.abs(PI @ Y.T - TAU).max() <= 5e-7 * scale, name
        Yrec = np.linalg.lstsq(PI, TAU, rcond=None)[0].T
        assert np.abs(Yrec - Y).max() <= 2e-5 * np.abs(Y).max(), (name, np.abs(Yrec - Y).max(), np.abs(Y).max())
