"""Host-side profile of one pipeline step at a BASELINE config (cProfile over a few IdentificationPipeline.run calls):
where the step time goes that the kernels do not account for.   usage: python tools/step_profile.py cfg4 [steps]"""
import cProfile, json, os, pstats, sys, time
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
if cfg == "cfg2":
    q, v, a = (rng.uniform(-6, 6, (N, 6)) for _ in range(3))
else:
    q, v, a = sample_inputs(robot.model, N, rng, 1.5, 2, 5)
pipe = IdentificationPipeline(robot, meta["param"], params_std=dict(zip(meta["names_std"], meta["phi_ref_raw"])),
                              coupling=meta["coupling"], chunk_samples=chunk)
pipe.set_samples(q, v, a)
pipe.set_tau_from_parameters(np.array([float(x) for x in meta["phi_ref_raw"]]), noise_std=0.0, seed=0)
pipe.run(); _lib.synchronize()
_lib.profile_enable(True, level=1); _lib.profile_reset()
for _ in range(steps):
    pipe.run()
_lib.synchronize()
print("kernel averages (HIP events):", {k: round(_lib.profile_get(k)[1] / max(_lib.profile_get(k)[0], 1), 3)
                                        for k in ("regressor_tree", "regressor_chain", "tsqr")})
_lib.profile_enable(False)
t0 = time.perf_counter()
pr = cProfile.Profile(); pr.enable()
for _ in range(steps):
    pipe.run()
_lib.synchronize()
pr.disable()
print("%.2f ms per step" % (1e3 * (time.perf_counter() - t0) / steps))
pstats.Stats(pr).sort_stats("tottime").print_stats(18)
