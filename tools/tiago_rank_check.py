"""Rank decision |R_ii| > 1e-8 on TIAGo at a size where the golden (N = 400) and the large-N answers differ: the device
TSQR against LAPACK (np.linalg.qr, mode='r') on the SAME reduced regressor, plus the pivot growth with N."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
from figaroh_plus_amd.pipeline import IdentificationPipeline
from figaroh_plus_amd.tools.robot import Robot
from gen_golden_inputs import sample_inputs  # noqa

meta = json.load(open(os.path.join(ROOT, "tests", "golden", "cfg3_tiago.json")))
g = np.load(os.path.join(ROOT, "tests", "golden", "cfg3_tiago.npz"))
robot = Robot.from_flat("tiago")
std = dict(zip(meta["names_std"], meta["phi_ref_raw"]))
gold = set(int(i) for i in g["idx_base"])
for N in (100_000, 400_000):
    q, v, a = sample_inputs(robot.model, N, np.random.default_rng(5), 1.5, 2, 5)
    pipe = IdentificationPipeline(robot, meta["param"], params_std=std, coupling=meta["coupling"])
    pipe.set_samples(q, v, a)
    out = pipe.run()
    extra = sorted(set(out["idx_base"]) - gold)
    d = np.asarray(out["absdiagR"])
    print("N=%d: device base params %d, beyond the golden set: %s with |Rii| %s" % (
        N, len(out["idx_base"]), extra, ["%.2e" % d[i] for i in extra]), flush=True)
    if N == 400_000:
        keep = [i for i in range(pipe.W.ref_cols) if i not in set(out["idx_e"])]  # reference numbering ...
        t0 = time.perf_counter()
        W = pipe.W.numpy()[:, pipe.device_columns(keep)]                           # ... read from the link-padded W
        R = np.linalg.qr(W, mode="r")
        dl = np.abs(np.diag(R))
        lap = [i for i in range(len(keep)) if dl[i] > 1e-8]
        print("LAPACK on the same %d x %d matrix (%.0f s): base params %d, identical index set: %s; "
              "max |device - LAPACK| over the pivots of the extra columns: %.1e (relative %.1e)" % (
                  W.shape[0], W.shape[1], time.perf_counter() - t0, len(lap), lap == list(out["idx_base"]),
                  max(abs(d[i] - dl[i]) for i in extra) if extra else 0.0,
                  max(abs(d[i] - dl[i]) / dl[i] for i in extra) if extra else 0.0), flush=True)
    del pipe
