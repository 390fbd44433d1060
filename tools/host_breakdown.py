import sys, time, json, os
sys.path.insert(0, os.getcwd())
import numpy as np
from figaroh_plus_amd import _lib, pipeline as P
from figaroh_plus_amd.pipeline import IdentificationPipeline
from figaroh_plus_amd.tools.robot import Robot
meta = json.load(open('tests/golden/cfg2_ur10.json'))
robot = Robot.from_flat('ur10'); param = meta['param']; std = dict(zip(meta['names_std'], meta['phi_ref_raw']))
N = 1000000
rng = np.random.default_rng(1); q, v, a = (rng.uniform(-6, 6, (N, 6)) for _ in range(3))
pipe = IdentificationPipeline(robot, param, params_std=std); pipe.set_samples(q, v, a)
pipe.set_tau_from_parameters(np.array([float(x) for x in meta['phi_ref_raw']]), noise_std=0.05)
for i in range(5): pipe.run()
marks = []
def wrap(obj, name):
    f = getattr(obj, name)
    def g(*a, **k):
        t0 = time.perf_counter(); r = f(*a, **k); marks.append((name, t0, time.perf_counter())); return r
    setattr(obj, name, g)
wrap(_lib, "regressor_build"); wrap(_lib, "tsqr_selected"); wrap(_lib, "regressor_tsqr_fused"); wrap(_lib, "select_columns")
wrap(IdentificationPipeline, "_finish")
lib = _lib.load()
orig_d2h = lib.figh_memcpy_d2h
class L:
    def __getattr__(self, k):
        f = getattr(lib, k)
        if k != "figh_memcpy_d2h": return f
        def g(*a):
            t0 = time.perf_counter(); r = f(*a); marks.append((k, t0, time.perf_counter())); return r
        return g
_lib.load = lambda: L()
acc = {}
tot = []
for i in range(30):
    marks.clear(); t0 = time.perf_counter(); pipe.run(); t1 = time.perf_counter(); tot.append(t1 - t0)
    prev = t0
    for name, a0, a1 in marks:
        acc.setdefault("gap before " + name, []).append(a0 - prev); acc.setdefault(name, []).append(a1 - a0); prev = a1
    acc.setdefault("after last", []).append(t1 - prev)
print("step median %.1f us" % (1e6 * np.median(tot)))
for k, v in acc.items(): print("  %-32s %.1f us" % (k, 1e6 * np.median(v)))
