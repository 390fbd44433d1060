import os, sys, json, time
import numpy as np
sys.path.insert(0, os.getcwd())
from figaroh_plus_amd import _lib
from figaroh_plus_amd.pipeline import IdentificationPipeline
from figaroh_plus_amd.tools.robot import Robot
meta = json.load(open('tests/golden/cfg2_ur10.json'))
robot = Robot.from_flat('ur10')
N = 1000000
rng = np.random.default_rng(1); q, v, a = (rng.uniform(-6, 6, (N, 6)) for _ in range(3))
pipe = IdentificationPipeline(robot, meta['param'], params_std=dict(zip(meta['names_std'], meta['phi_ref_raw'])))
pipe.set_samples(q, v, a)
pipe.set_tau_from_parameters(np.array([float(x) for x in meta['phi_ref_raw']]), noise_std=0.05)
for _ in range(5): pipe.run()
_lib.profile_enable(True, level=1); _lib.profile_reset()
ts = []
for _ in range(40):
    t0 = time.perf_counter(); pipe.run(); ts.append(time.perf_counter() - t0)
cnt, ms = _lib.profile_get("fused_chain_tsqr")
print(os.path.basename(_lib.LIB_PATH), "step median %.4f min %.4f mean %.4f ms; fused kernel avg %.4f ms" % (1e3*np.median(ts), 1e3*min(ts), 1e3*np.mean(ts), ms/cnt))
