"""BASELINE.json configs 3-5 at their full sizes through the HBM-resident pipeline (TIAGo 1e6, TALOS 4e6) and the
streamed one (human 1e7: W would be 269 GB): wall time per pass, kernel times, structural result vs the golden
fixtures (which were produced by the reference's code at N = 32 / 400 samples)."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
from figaroh_plus_amd import _lib
from figaroh_plus_amd.pipeline import IdentificationPipeline
from figaroh_plus_amd.tools.robot import Robot
from figaroh_plus_amd.tools.randomdata import sample_inputs  # noqa

which = sys.argv[1:] or ["cfg3_tiago", "cfg4_talos", "cfg5_human"]
SIZES = {"cfg3_tiago": ("tiago", 1_000_000, None), "cfg4_talos": ("talos", 4_000_000, None),
         "cfg5_human": ("human", 10_000_000, 500_000)}
for cfg in which:
    mn, N, chunk = SIZES[cfg]
    if os.environ.get("FIGH_FULL_SCALE"):
        N = int(N * float(os.environ["FIGH_FULL_SCALE"]))
    meta = json.load(open(os.path.join(ROOT, "tests", "golden", cfg + ".json")))
    g = np.load(os.path.join(ROOT, "tests", "golden", cfg + ".npz"))
    robot = Robot.from_flat(mn)
    t0 = time.perf_counter()
    q, v, a = sample_inputs(robot.model, N, np.random.default_rng(5), 1.5, 2, 5)
    t_gen = time.perf_counter() - t0
    std = dict(zip(meta["names_std"], meta["phi_ref_raw"]))
    pipe = IdentificationPipeline(robot, meta["param"], params_std=std, coupling=meta["coupling"], chunk_samples=chunk)
    pipe.set_samples(q, v, a)
    del q, v, a
    pipe.set_tau_from_parameters(np.array([float(x) for x in meta["phi_ref_raw"]]))
    out = pipe.run()
    _lib.profile_enable(True); _lib.profile_reset()
    t0 = time.perf_counter(); out = pipe.run(); dt = time.perf_counter() - t0
    prof = {k: round(_lib.profile_get(k)[1], 1) for k in ("regressor_tree", "colsq", "tsqr", "tsqr_reduce", "tsqr_small")}
    _lib.profile_enable(False)
    same_e = out["idx_e"] == list(g["idx_e"]); same_b = out["idx_base"] == list(g["idx_base"])
    dep = sorted([x for i, x in enumerate(out["absdiagR"]) if i not in set(out["idx_base"])])
    phi_err = (np.abs(out["phi_ls"] - g["phi_from_std"]).max() / np.abs(g["phi_from_std"]).max()) if same_b else float("nan")
    rps = 24 if mn == "tiago" else 6
    ncols = len(meta["names_std"])
    print("%-11s N=%d (W %.1f GB%s)  pass %.1f ms = %.2e samples/s  idx_e ok %s  base params %d (golden %d, identical %s)  "
          "largest dependent |Rii| %.1e  phi err %.1e  kernels(ms) %s  [inputs generated in %.0f s]" % (
              cfg, N, rps * N * ncols * 8 / 1e9, ", streamed in chunks of %d" % chunk if chunk else "", dt * 1e3, N / dt,
              same_e, len(out["idx_base"]), len(g["idx_base"]), same_b, dep[-1] if dep else 0.0, phi_err, prof, t_gen), flush=True)
    del pipe, out
