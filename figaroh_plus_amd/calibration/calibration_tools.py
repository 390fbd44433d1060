"""The one piece of ``src/figaroh/calibration/calibration_tools.py`` that runs on the hot path's kernels
(SURVEY section 8f-4): ``calculate_base_kinematics_regressor`` (``:1469-1561``) -- column elimination, independent
column selection and base regressor of the KINEMATIC regressor, i.e. the second caller of
``eliminate_non_dynaffect`` / ``get_baseIndex`` / ``get_baseParams`` / ``build_baseRegressor``.

The kinematic regressor itself (``calculate_identifiable_kinematics_model``, ``:1391-1466``: Pinocchio frame
Jacobians and placement regressors over calibration poses) belongs to the geometric-calibration subsystem, which is
out of scope (component 14); the caller passes it in as ``kinematics_model`` -- with Pinocchio
installed that is the reference's own function.  Parameter-name helpers ``get_geo_offset`` (``:301-331``) and
``get_joint_offset`` (``:252-298``) are mirrored because the base-parameter names are built from them.
"""
import numpy as np

from ..tools.qrdecomposition import build_baseRegressor, get_baseIndex, get_baseParams
from ..tools.regressor import eliminate_non_dynaffect

TOL_QR = 1e-8


def get_joint_offset(model, joint_names):
    """{"offsetRZ_<joint>": 0, ...} -- one entry per joint degree of freedom, named after the joint model with
    "JointModel" replaced by "offset"; the second and later dofs of a joint carry their 1-based number
    (calibration_tools.py:252-298, including its special case for the model called "canopies").  Like the reference,
    the names come from ``model.names``, not from the argument."""
    names, joints = list(model.names[1:]), list(model.joints[1:])
    assert len(names) == len(joints), "Number of jointnames does not match number of joints! Please check\
        imported model."
    keys = []
    for name, joint in zip(names, joints):
        kind = joint.shortname()
        if model.name == "canopies" and "RevoluteUnaligned" in kind:
            kind = kind.replace("RevoluteUnaligned", "RZ")
        stem = kind.replace("JointModel", "offset")
        keys.extend((stem if dof == 0 else "%s%d" % (stem, dof + 1)) + "_" + name for dof in range(joint.nv))
    return dict.fromkeys(keys, 0)


def get_geo_offset(joint_names):
    """{"d_px_<joint>": 0, "d_py_<joint>": 0, ... "d_phiz_<joint>": 0} per joint (calibration_tools.py:301-331)."""
    return dict.fromkeys(("%s_%s" % (p, j) for j in joint_names for p in
                          ("d_px", "d_py", "d_pz", "d_phix", "d_phiy", "d_phiz")), 0)


def calculate_base_kinematics_regressor(q, model, data, param, tol_qr=TOL_QR, kinematics_model=None):
    """(Rrand_b, R_b, R_e, paramsrand_base, paramsrand_e) -- calibration_tools.py:1469-1561 from the point where the
    kinematic regressors exist; the base parameter names are appended to ``param["param_name"]`` and the three shapes
    are printed, as the reference does.  ``kinematics_model(q, model, data, param)`` stands for the reference's
    ``calculate_identifiable_kinematics_model``; the four regressor reductions run on the device."""
    if kinematics_model is None:
        raise NotImplementedError("pass kinematics_model=calculate_identifiable_kinematics_model: the kinematic "
                                  "regressor is part of the calibration subsystem (needs Pinocchio), not of this path")
    names = list(model.names[1:])
    candidates = {"joint_offset": get_joint_offset(model, names), "full_params": get_geo_offset(names)}
    # regressor over random configurations (fixed base: the model draws them itself, hence []), and over the given
    # ones when there are any
    R_random = kinematics_model(q if param["free_flyer"] else [], model, data, param)
    R_given = kinematics_model(q, model, data, param) if np.any(np.array(q)) else R_random
    if param["calib_model"] not in candidates:  # the reference leaves geo_params_sel unbound
        raise UnboundLocalError("local variable 'geo_params_sel' referenced before assignment")
    selected = candidates[param["calib_model"]]
    # random data decide which parameters are identifiable and which of them are independent ...
    Rrand_e, paramsrand_e = eliminate_non_dynaffect(R_random, selected, tol_e=1e-6)
    idx_base = get_baseIndex(Rrand_e, paramsrand_e, tol_qr=tol_qr)
    Rrand_b, paramsrand_base, _ = get_baseParams(Rrand_e, paramsrand_e, tol_qr=tol_qr)
    # ... the given data are reduced the same way (their own base set is computed and dropped, as in the reference)
    # and restricted to the columns chosen on the random data
    R_e, params_e = eliminate_non_dynaffect(R_given, selected, tol_e=1e-6)
    get_baseParams(R_e, params_e, tol_qr=tol_qr)
    R_b = build_baseRegressor(R_e, idx_base)
    param["param_name"].extend(paramsrand_e[j] for j in idx_base)
    print("shape of full regressor, reduced regressor, base regressor: ", R_random.shape, Rrand_e.shape, Rrand_b.shape)
    return Rrand_b, R_b, R_e, paramsrand_base, paramsrand_e
