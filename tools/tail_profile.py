"""cProfile of IdentificationPipeline.run for a wide config (host tail hunting)."""
import cProfile, json, os, pstats, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
from figaroh_plus_amd.pipeline import IdentificationPipeline
from figaroh_plus_amd.tools.robot import Robot
from gen_golden_inputs import sample_inputs
cfg, mn = (sys.argv[1], sys.argv[2]) if len(sys.argv) > 2 else ("cfg3_tiago", "tiago")
meta = json.load(open(os.path.join(ROOT, "tests", "golden", cfg + ".json")))
robot = Robot.from_flat(mn)
q, v, a = sample_inputs(robot.model, 50000, np.random.default_rng(5), 1.5, 2, 5)
std = dict(zip(meta["names_std"], meta["phi_ref_raw"]))
pipe = IdentificationPipeline(robot, meta["param"], params_std=std, coupling=meta["coupling"])
pipe.set_samples(q, v, a); pipe.set_tau_from_parameters(np.array([float(x) for x in meta["phi_ref_raw"]]))
pipe.run()
pr = cProfile.Profile(); pr.enable(); pipe.run(); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
