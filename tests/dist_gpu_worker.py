"""One rank of the 2-process pipeline test (tests/test_gpu_parity.py::test_two_process_pipeline_on_one_device): runs
IdentificationPipeline on its shard of the UR10 golden samples with the exchange that dist.exchange_from_env builds, and
writes what it got.  Started as a fresh process per rank (RANK / WORLD_SIZE / MASTER_* in the environment)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

from figaroh_plus_amd import _lib  # noqa: E402
from figaroh_plus_amd.dist import exchange_from_env, shard_range  # noqa: E402
from figaroh_plus_amd.pipeline import IdentificationPipeline  # noqa: E402
from figaroh_plus_amd.tools.robot import Robot  # noqa: E402


def main(out_path):
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    _lib.check(_lib.load().figh_device_set(0))  # both ranks drive the one device of the test box
    # the ranks report the SAME device: RCCL (one GPU per rank) is ruled out in the collective preflight and every rank
    # takes the host-staged exchange -- the decision path of a mis-launched job, exercised on purpose
    ex, info = exchange_from_env("rccl", device_key=("testbox", 0))
    with open(os.path.join(ROOT, "tests", "golden", "cfg2_ur10.json")) as f:
        meta = json.load(f)
    g = np.load(os.path.join(ROOT, "tests", "golden", "cfg2_ur10.npz"))
    q, v, a, tau = g["q_big"], g["v_big"], g["a_big"], g["tau"]
    N = len(q)
    lo, hi = shard_range(N, rank, world)
    tau_shard = np.ascontiguousarray(tau.reshape(6, N)[:, lo:hi]).reshape(-1)  # rows j*N + i -> j*(hi-lo) + (i-lo)
    robot = Robot.from_flat("ur10")
    pipe = IdentificationPipeline(robot, meta["param"], params_std=dict(zip(meta["names_std"], meta["phi_ref_raw"])),
                                  exchange=ex)
    pipe.set_samples(q[lo:hi], v[lo:hi], a[lo:hi], tau_shard)
    out = pipe.run()
    with open(out_path, "w") as f:
        json.dump({"rank": rank, "collective": info["collective"], "idx_e": out["idx_e"], "idx_base": out["idx_base"],
                   "params_base": out["params_base"], "phi_ls": out["phi_ls"].tolist(), "rows": out["rows"],
                   "col_norm": out["col_norm"].tolist()}, f)
    ex.barrier()


if __name__ == "__main__":
    main(sys.argv[1])
