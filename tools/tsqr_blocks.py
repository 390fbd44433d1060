"""Per joint-block in-kernel profile of the level-0 TSQR (run with FIGH_TSQR_DBG=4): ticks per column step for the
rows of each joint separately (block j has its own mix of live chunks per step)."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from figaroh_plus_amd import _lib
from figaroh_plus_amd.pipeline import IdentificationPipeline, _View
from figaroh_plus_amd.tools.robot import Robot
meta = json.load(open(ROOT + '/tests/golden/cfg2_ur10.json'))
robot = Robot.from_flat('ur10'); param = meta['param']; std = dict(zip(meta['names_std'], meta['phi_ref_raw']))
N = 1000000
rng = np.random.default_rng(1); q, v, a = (rng.uniform(-6, 6, (N, 6)) for _ in range(3))
pipe = IdentificationPipeline(robot, param, params_std=std); pipe.set_samples(q, v, a)
pipe.set_tau_from_parameters(np.array([float(x) for x in meta['phi_ref_raw']]), noise_std=0.05)
out = pipe.run()
keep = [i for i in range(84) if i not in set(out["idx_e"])]
d_idx = _lib.DeviceArray.from_host(np.asarray(keep, dtype=np.int32))
d_R = _lib.DeviceArray((50 * 50,))
for j in range(6):
    print("joint block", j + 1, flush=True)
    Wv = _View(pipe.W.buf, j * N * 84 * 8)
    tv = _View(pipe.d_tau, j * N * 8)
    for _ in range(2):
        _lib.tsqr(Wv, N, 84, d_idx, 49, tv, None, d_R)
    _lib.synchronize()
