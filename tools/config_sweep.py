"""All five BASELINE configs through the HBM-resident pipeline at a moderate N: wall time per stage and the
structural result (eliminated / base columns) against the golden fixtures."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
from figaroh_plus_amd import _lib
from figaroh_plus_amd.pipeline import IdentificationPipeline
from figaroh_plus_amd.tools.robot import Robot
from gen_golden_inputs import sample_inputs  # noqa

N = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
for cfg, mn in (("cfg1_tx40", "tx40"), ("cfg2_ur10", "ur10"), ("cfg3_tiago", "tiago"), ("cfg4_talos", "talos"), ("cfg5_human", "human")):
    meta = json.load(open(os.path.join(ROOT, "tests", "golden", cfg + ".json")))
    g = np.load(os.path.join(ROOT, "tests", "golden", cfg + ".npz"))
    robot = Robot.from_flat(mn)
    rng = np.random.default_rng(5)
    q, v, a = sample_inputs(robot.model, N, rng, *( (6, 10, 30) if mn == "tx40" else (6, 6, 6) if mn == "ur10" else (1.5, 2, 5)))
    std = dict(zip(meta["names_std"], meta["phi_ref_raw"]))
    pipe = IdentificationPipeline(robot, meta["param"], params_std=std, coupling=meta["coupling"])
    pipe.set_samples(q, v, a)
    pipe.set_tau_from_parameters(np.array([float(x) for x in meta["phi_ref_raw"]]))
    out = pipe.run()
    _lib.profile_enable(True); _lib.profile_reset()
    t0 = time.perf_counter(); out = pipe.run(); dt = time.perf_counter() - t0
    prof = {k: round(_lib.profile_get(k)[1], 2) for k in ("regressor_chain", "regressor_tree", "colsq", "tsqr", "tsqr_tree", "tsqr_reduce", "tsqr_small")}
    _lib.profile_enable(False)
    same_e = out["idx_e"] == list(g["idx_e"]); same_b = out["idx_base"] == list(g["idx_base"])
    dep = [x for i, x in enumerate(out["absdiagR"]) if i not in set(out["idx_base"])]
    print("%-11s N=%d W %dx%d  step %.1f ms (%.2e samples/s)  idx_e ok %s  idx_base ok %s (%d)  max dep |Rii| %.1e  phi err %.1e  kernels(ms) %s" % (
        cfg, N, pipe.W.rows, pipe.W.ref_cols, dt * 1e3, N / dt, same_e, same_b, len(out["idx_base"]), max(dep) if dep else 0,
        (np.abs(out["phi_ls"] - g["phi_from_std"]).max() / np.abs(g["phi_from_std"]).max()) if same_b else float("nan"),
        prof), flush=True)
    del pipe
