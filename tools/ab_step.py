"""UR10 1e6-sample pass with the library at FIGH_LIB_PATH (default: in-tree): step time (wall clock, 40 passes) and the
library's own event times per kernel family.  usage: [FIGH_LIB_PATH=ab/libfigh_prev.so] python tools/ab_step.py"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from figaroh_plus_amd import _lib
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
for _ in range(5):
    out = pipe.run()
_lib.synchronize(); t0 = time.perf_counter()
for _ in range(40):
    out = pipe.run()
_lib.synchronize(); step = (time.perf_counter() - t0) / 40
_lib.profile_enable(True, level=2); _lib.profile_reset()
for _ in range(10):
    pipe.run()
parts = []
for k in ("regressor_chain", "tsqr", "tsqr_tree", "select_columns"):
    c, ms = _lib.profile_get(k)
    parts.append("%s %.4f" % (k, ms / max(c, 1)))
print("%-18s step %.4f ms | %s | base %d" % (os.path.basename(_lib.LIB_PATH), 1e3 * step, " ".join(parts), len(out["idx_base"])))
