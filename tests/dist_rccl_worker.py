"""One rank of the TWO-DEVICE RCCL tests (tests/test_gpu_parity.py::test_rccl_two_devices_*): a fresh process per rank, rank r on
device r, RANK / WORLD_SIZE / MASTER_* in the environment, control plane = dist.SocketGroup (no PyTorch).  The exchange must
come out of dist.exchange_from_env as RCCL (two distinct devices); the primitives and one sharded IdentificationPipeline pass
per model run through it, and what this rank got is written to ``argv[1]`` for the parent test to compare with the golden
vectors."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402

from figaroh_plus_amd import _lib  # noqa: E402
from figaroh_plus_amd.dist import allreduce_normal_terms, exchange_from_env, shard_range  # noqa: E402
from figaroh_plus_amd.pipeline import IdentificationPipeline  # noqa: E402


def main(out_path, models):
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    lib = _lib.load()
    _lib.check(lib.figh_device_set(int(os.environ["LOCAL_RANK"]) % _lib.device_count()))
    ex, info = exchange_from_env("rccl", rendezvous="socket")
    res = {"rank": rank, "collective": info["collective"], "exchange_class": type(ex).__name__}
    # --- primitives on device buffers
    n = 84
    mine = np.arange(n, dtype=np.float64) * (rank + 1) + 0.25 * rank
    d = _lib.DeviceArray.from_host(mine)
    ex.sum_columns_device(d, n)
    res["sum_columns_device"] = d.to_host().tolist()
    d2 = _lib.DeviceArray.from_host(mine)
    res["sum_columns"] = np.asarray(ex.sum_columns(d2, n)).tolist()
    nc = 7
    tri = np.triu(np.arange(nc * nc, dtype=np.float64).reshape(nc, nc) + 100.0 * rank)
    d_tri = _lib.DeviceArray.from_host(tri.reshape(-1))
    stack, count = ex.stack_triangles(d_tri, nc)
    host = np.empty(count * nc * nc)
    _lib.check(lib.figh_memcpy_d2h(host.ctypes.data, stack.ptr, host.nbytes))
    res["stack"] = host.tolist()
    res["stack_count"] = int(count)
    rng = np.random.default_rng(40 + rank)
    A = rng.standard_normal((30 + rank, 5))
    t = rng.standard_normal(30 + rank)
    cs, G, g, tsq, rows = allreduce_normal_terms(ex, (A * A).sum(0), A.T @ A, A.T @ t, float(t @ t), len(t))
    res["normal_terms"] = {"colsq": cs.tolist(), "G": G.tolist(), "g": g.tolist(), "tau_sq": tsq, "rows": rows}
    # --- one sharded pass of the identification pipeline per model
    from conftest import Golden
    from figaroh_plus_amd.tools.randomdata import sample_inputs
    res["models"] = {}
    for cfg in models:
        gold = Golden(cfg)
        robot = gold.robot()
        if cfg == "cfg2_ur10":  # (enough samples per rank for the fused launch: 4096)
            q, v, a = (np.random.default_rng(6).uniform(-6, 6, (3, 20000 + 36, 6)))
        else:
            q, v, a = sample_inputs(robot.model, 30000 + 11, np.random.default_rng(5), 1.5, 2, 5)  # same on every rank
        N = len(q)
        lo, hi = shard_range(N, rank, world)
        pipe = IdentificationPipeline(robot, gold.param, params_std=gold.params_std(), coupling=gold.coupling, exchange=ex)
        pipe.set_samples(q[lo:hi], v[lo:hi], a[lo:hi])
        pipe.set_tau_from_parameters(gold.phi_ref())
        out = pipe.run()
        out = pipe.run()  # (second pass: the speculative / fused forms)
        res["models"][cfg] = {"idx_e": out["idx_e"], "idx_base": out["idx_base"], "params_base": out["params_base"],
                              "phi_ls": out["phi_ls"].tolist(), "rows": int(out["rows"]), "col_norm": out["col_norm"].tolist(),
                              "absdiagR": out["absdiagR"].tolist(), "fused_passes": int(pipe.fused_passes)}
        del pipe
    with open(out_path, "w") as f:
        json.dump(res, f)
    ex.control.barrier()
    if hasattr(ex, "close"):
        ex.close()


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2:])
