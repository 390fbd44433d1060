"""Level-0 narrow TSQR (nc <= 80) A/B on the UR10 problem: time of figh_tsqr with the structure hint, per library build.
usage: FIGH_LIB_PATH=... python tools/narrow_ab.py"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from figaroh_plus_amd import _lib
if os.environ.get("FIGH_OLD_ABI"):  # a round-1 build: symbols added since are not there
    for k in ("figh_regressor_build_padded", "figh_comm_available", "figh_place_block", "figh_host_wait_mode"):
        _lib.SIGNATURES.pop(k, None)
from figaroh_plus_amd.pipeline import IdentificationPipeline
from figaroh_plus_amd.tools.robot import Robot
meta = json.load(open(os.path.join(ROOT, "tests", "golden", "cfg2_ur10.json")))
robot = Robot.from_flat("ur10")
N = 1_000_000
rng = np.random.default_rng(3)
q, v, a = (rng.uniform(-6, 6, (N, 6)) for _ in range(3))
pipe = IdentificationPipeline(robot, meta["param"], params_std=dict(zip(meta["names_std"], meta["phi_ref_raw"])))
pipe.set_samples(q, v, a)
pipe.set_tau_from_parameters(np.array([float(x) for x in meta["phi_ref_raw"]]), noise_std=0.05)
for _ in range(3):
    pipe.run()
_lib.profile_enable(True); _lib.profile_reset()
for _ in range(10):
    pipe.run()
for k in ("regressor_chain", "tsqr", "tsqr_reduce", "tsqr_small"):
    c, ms = _lib.profile_get(k)
    print(os.path.basename(_lib.LIB_PATH), k, c, "avg %.4f ms" % (ms / max(c, 1)))
