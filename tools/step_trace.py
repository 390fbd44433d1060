"""Timestamps of every C-ABI call of a few pipeline steps (host clock): which call, or which gap between calls, holds the
time that the kernels do not account for.   usage: python tools/step_trace.py cfg4 [steps]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from figaroh_plus_amd import _lib
from figaroh_plus_amd.pipeline import IdentificationPipeline
from figaroh_plus_amd.tools.randomdata import sample_inputs
from figaroh_plus_amd.tools.robot import Robot
import bench

cfg = sys.argv[1] if len(sys.argv) > 1 else "cfg4"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
fixture, model_name, N, chunk = bench.CONFIGS[cfg]
meta = json.load(open(os.path.join(ROOT, "tests", "golden", fixture + ".json")))
robot = Robot.from_flat(model_name)
rng = np.random.default_rng(1)
q, v, a = sample_inputs(robot.model, N, rng, 1.5, 2, 5)
pipe = IdentificationPipeline(robot, meta["param"], params_std=dict(zip(meta["names_std"], meta["phi_ref_raw"])),
                              coupling=meta["coupling"], chunk_samples=chunk)
pipe.set_samples(q, v, a)
pipe.set_tau_from_parameters(np.array([float(x) for x in meta["phi_ref_raw"]]), noise_std=0.0, seed=0)
if len(sys.argv) > 3:
    _lib.profile_enable(True, level=int(sys.argv[3]))
pipe.run(); pipe.run(); _lib.synchronize()
lib = _lib.load()
log = []
for name in _lib.SIGNATURES:
    f = getattr(lib, name)
    def wrap(*args, _f=f, _n=name):
        t0 = time.perf_counter(); r = _f(*args); log.append((_n, t0, time.perf_counter())); return r
    setattr(lib, name, wrap)
T0 = time.perf_counter()
for _ in range(steps):
    log.append(("== step", time.perf_counter(), time.perf_counter()))
    pipe.run()
prev = T0
for name, t0, t1 in log:
    print("%9.3f ms  +gap %8.3f  call %8.3f  %s" % (1e3 * (t0 - T0), 1e3 * (t0 - prev), 1e3 * (t1 - t0), name))
    prev = t1
