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
    """{"offsetRZ_<joint>": 0, ...}: one entry per joint degree of freedom, named after the joint model
    (calibration_tools.py:252-298; a multi-dof joint numbers its entries from the second one on)."""
    joint_off = []
    joint_names = list(model.names[1:])
    joints = list(model.joints[1:])
    assert len(joint_names) == len(joints), "Number of jointnames does not match number of joints! Please check\
        imported model."
    for id, joint in enumerate(joints):
        name = joint_names[id]
        shortname = joint.shortname()
        if model.name == "canopies":
            if "RevoluteUnaligned" in shortname:
                shortname = shortname.replace("RevoluteUnaligned", "RZ")
        for i in range(joint.nv):
            if i > 0:
                offset_param = shortname.replace("JointModel", "offset") + "{}".format(i + 1) + "_" + name
            else:
                offset_param = shortname.replace("JointModel", "offset") + "_" + name
            joint_off.append(offset_param)
    return dict(zip(joint_off, [0] * len(joint_off)))


def get_geo_offset(joint_names):
    """{"d_px_<joint>": 0, "d_py_<joint>": 0, ... "d_phiz_<joint>": 0} per joint (calibration_tools.py:301-331)."""
    tpl_names = ["d_px", "d_py", "d_pz", "d_phix", "d_phiy", "d_phiz"]
    geo_params = [j + "_" + joint_names[i] for i in range(len(joint_names)) for j in tpl_names]
    return dict(zip(geo_params, [0] * len(geo_params)))


def calculate_base_kinematics_regressor(q, model, data, param, tol_qr=TOL_QR, kinematics_model=None):
    """(Rrand_b, R_b, R_e, paramsrand_base, paramsrand_e) -- calibration_tools.py:1469-1561, statement by statement
    from the point where the kinematic regressors exist; appends the base parameter names to ``param["param_name"]``
    like the reference.  ``kinematics_model(q, model, data, param)`` stands for the reference's
    ``calculate_identifiable_kinematics_model``; the four regressor reductions run on the device."""
    if kinematics_model is None:
        raise NotImplementedError("pass kinematics_model=calculate_identifiable_kinematics_model: the kinematic "
                                  "regressor is part of the calibration subsystem (needs Pinocchio), not of this path")
    joint_names = [name for i, name in enumerate(model.names[1:])]
    geo_params = get_geo_offset(joint_names)
    joint_offsets = get_joint_offset(model, joint_names)
    # calculate kinematic regressor with random configs
    if not param["free_flyer"]:
        Rrand = kinematics_model([], model, data, param)
    else:
        Rrand = kinematics_model(q, model, data, param)
    # calculate kinematic regressor with input configs
    if np.any(np.array(q)):
        R = kinematics_model(q, model, data, param)
    else:
        R = Rrand
    if param["calib_model"] == "joint_offset":
        geo_params_sel = joint_offsets
    elif param["calib_model"] == "full_params":
        geo_params_sel = geo_params
    else:  # the reference leaves geo_params_sel unbound
        raise UnboundLocalError("local variable 'geo_params_sel' referenced before assignment")
    Rrand_sel, R_sel = Rrand, R
    # remove non affect columns from random data => reduced regressor
    Rrand_e, paramsrand_e = eliminate_non_dynaffect(Rrand_sel, geo_params_sel, tol_e=1e-6)
    # indices of independent columns (base param) w.r.t to reduced regressor
    idx_base = get_baseIndex(Rrand_e, paramsrand_e, tol_qr=tol_qr)
    # get base regressor and base params from random data
    Rrand_b, paramsrand_base, _ = get_baseParams(Rrand_e, paramsrand_e, tol_qr=tol_qr)
    # remove non affect columns from GIVEN data
    R_e, params_e = eliminate_non_dynaffect(R_sel, geo_params_sel, tol_e=1e-6)
    # get base param from given data
    R_gb, params_gbase, _ = get_baseParams(R_e, params_e, tol_qr=tol_qr)
    # get base regressor from GIVEN data
    R_b = build_baseRegressor(R_e, idx_base)
    # update calibrating param['param_name']/calibrating parameters
    for j in idx_base:
        param["param_name"].append(paramsrand_e[j])
    print("shape of full regressor, reduced regressor, base regressor: ", Rrand.shape, Rrand_e.shape, Rrand_b.shape)
    return Rrand_b, R_b, R_e, paramsrand_base, paramsrand_e
