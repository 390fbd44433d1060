"""Fixture for set_missing_params_setting (src/figaroh/identification/identification_tools.py:86-165): the reference's own
function on model stand-ins that carry the four limit arrays, in the cases its branches distinguish (limits present /
zero velocity limits; friction and external-wrench offsets on and off).  Run in the authoring container (needs
/root/reference); writes tests/golden/missing_params.json."""
import contextlib
import io
import json
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)
import gen_golden as gg  # noqa: E402  (loads the reference modules by path)


def cases():
    rng = np.random.default_rng(3)
    for name, nq, nv, zero_vel, fric, offs in (("limits_present", 6, 6, False, True, False),
                                              ("no_velocity_limits", 7, 7, True, False, True),
                                              ("floating_base", 9, 8, True, True, True)):
        yield name, {
            "lower": (-rng.uniform(1, 3, nq)).tolist(), "upper": rng.uniform(1, 3, nq).tolist(),
            "velocity": (np.zeros(nq) if zero_vel else rng.uniform(1, 2, nq)).tolist(),
            "effort": rng.uniform(10, 20, nq).tolist(), "nq": nq, "nv": nv,
            "settings": {"q_lim_def": 1.57, "dq_lim_def": 5.0, "tau_lim_def": 4.0, "ddq_lim_def": 20.0,
                         "has_friction": fric, "external_wrench_offsets": offs},
        }


def main():
    out = {}
    for name, c in cases():
        model = types.SimpleNamespace(lowerPositionLimit=np.array(c["lower"]), upperPositionLimit=np.array(c["upper"]),
                                      velocityLimit=np.array(c["velocity"]), effortLimit=np.array(c["effort"]),
                                      nq=c["nq"], nv=c["nv"])
        robot = types.SimpleNamespace(model=model)
        buf = io.StringIO()
        with contextlib.redirect_stdout(buf):
            res = gg.ref_idt.set_missing_params_setting(robot, dict(c["settings"]))
        out[name] = {"input": c, "printed": buf.getvalue(),
                     "result": {k: (v.tolist() if isinstance(v, np.ndarray) else v) for k, v in res.items()},
                     "model_after": {"lower": model.lowerPositionLimit.tolist(), "upper": model.upperPositionLimit.tolist(),
                                     "velocity": model.velocityLimit.tolist(), "effort": model.effortLimit.tolist()}}
    with open(os.path.join(ROOT, "tests", "golden", "missing_params.json"), "w") as f:
        json.dump(out, f, indent=1)
    print("wrote", len(out), "cases")


if __name__ == "__main__":
    main()
