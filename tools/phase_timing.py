"""Wall-clock breakdown of IdentificationPipeline.run on the GPU box (host overhead hunting)."""
import sys, time, json, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from figaroh_plus_amd import _lib
from figaroh_plus_amd.pipeline import IdentificationPipeline
from figaroh_plus_amd.tools.robot import Robot
meta = json.load(open(ROOT + '/tests/golden/cfg2_ur10.json'))
robot = Robot.from_flat('ur10'); param = meta['param']; std = dict(zip(meta['names_std'], meta['phi_ref_raw']))
N = 1000000
rng = np.random.default_rng(1); q, v, a = (rng.uniform(-6, 6, (N, 6)) for _ in range(3))
pipe = IdentificationPipeline(robot, param, params_std=std); pipe.set_samples(q, v, a)
pipe.set_tau_from_parameters(np.array([float(x) for x in meta['phi_ref_raw']]), noise_std=0.05)
for i in range(3): pipe.run()
marks = []
orig = {}
def wrap(mod, name):
    f = getattr(mod, name)
    def g(*a, **k):
        t0 = time.perf_counter(); r = f(*a, **k); marks.append((name, time.perf_counter() - t0)); return r
    setattr(mod, name, g)
for nm in ("regressor_build", "tsqr", "gather_cols", "tsqr_merge"): wrap(_lib, nm)
lib = _lib.load()
class L:
    def __getattr__(self, k):
        f = getattr(lib, k)
        def g(*a):
            t0 = time.perf_counter(); r = f(*a); marks.append((k, time.perf_counter() - t0)); return r
        return g
_lib.load = lambda: L()
tot = []
for i in range(10):
    marks.clear(); t0 = time.perf_counter(); pipe.run(); tot.append(time.perf_counter() - t0)
    if i == 9:
        acc = sum(m[1] for m in marks)
        print("step %.3f ms, inside instrumented calls %.3f ms" % (tot[-1] * 1e3, acc * 1e3))
        for k, t in marks: print("  %-22s %.3f ms" % (k, t * 1e3))
print("steps ms", [round(1e3 * t, 2) for t in tot])
