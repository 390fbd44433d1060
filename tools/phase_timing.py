import sys, time, json, os
sys.path.insert(0, '/root/repo')
import numpy as np
from figaroh_plus_amd import _lib
from figaroh_plus_amd.pipeline import IdentificationPipeline
from figaroh_plus_amd.tools.robot import Robot
ROOT='/root/repo'
meta=json.load(open(ROOT+'/tests/golden/cfg2_ur10.json'))
robot=Robot.from_flat('ur10'); param=meta['param']; std=dict(zip(meta['names_std'],meta['phi_ref_raw']))
N=1000000
rng=np.random.default_rng(1); q,v,a=(rng.uniform(-6,6,(N,6)) for _ in range(3))
pipe=IdentificationPipeline(robot,param,params_std=std); pipe.set_samples(q,v,a)
pipe.set_tau_from_parameters(np.array([float(x) for x in meta['phi_ref_raw']]),noise_std=0.05)
for i in range(3): pipe.run()
_lib.synchronize()
ts=[]
for i in range(20):
    t0=time.perf_counter(); pipe.run(); _lib.synchronize(); ts.append(time.perf_counter()-t0)
print("per-step ms:", [round(1e3*t,2) for t in ts])
import cProfile, pstats
pr=cProfile.Profile(); pr.enable()
for i in range(5): pipe.run()
pr.disable(); pstats.Stats(pr).sort_stats('cumulative').print_stats(18)
_lib.profile_enable(True); _lib.profile_reset()
ts=[]
for i in range(20):
    t0=time.perf_counter(); pipe.run(); _lib.synchronize(); ts.append(time.perf_counter()-t0)
print("profiled per-step ms:", [round(1e3*t,2) for t in ts])
print(_lib.profile_get("tsqr"), _lib.profile_get("regressor_chain"))
