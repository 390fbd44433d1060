#!/usr/bin/env python3
"""bench.py -- samples/s of regressor build + TSQR base-parameter solve on MI355X (BASELINE.json metric).

One "step" = one pass of the hot path over this rank's resident batch of synthetic (q, qd, qdd):
K1 regressor assembly (W materialised in the reference layout, column norms fused) -> elimination ->
K3 Householder TSQR over the kept columns + tau -> host tail (rank decision, regrouping, beta, strings,
phi) -> [N>1: RCCL all-reduce of column norms + all-gather of R factors].  Inputs are resident in HBM
before the timed region.  Workload = BASELINE.json configs[1] (UR10 6-DoF, 1e6 samples per GPU: weak scaling).

  python bench.py [--gpus N] [--steps K] [--warmup W] [--samples S] [--no-cpu-baseline]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FP64_PEAK_TFLOPS = 78.6   # MI355X datasheet: fp64 vector == fp64 matrix (v_mfma_f64_16x16x4) peak
HBM_PEAK_GBS = 8000.0     # /opt/skills/guides/MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)


def cpu_baseline(flat, seed, n_cpu):
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import cpu_baseline as cb  # oracle: measured here as the reported baseline, never shipped

    rng = np.random.default_rng(seed)
    q, v, a = (rng.uniform(-6, 6, (n_cpu, 6)) for _ in range(3))
    tau = rng.standard_normal(6 * n_cpu)
    t0 = time.perf_counter()
    _, stages = cb.faithful_pass(flat, q, v, a, tau)
    dt = time.perf_counter() - t0
    try:
        from threadpoolctl import threadpool_info
        blas = max([p.get("num_threads", 1) for p in threadpool_info()] + [1])
    except Exception:
        blas = os.cpu_count()
    return {
        "value": n_cpu / dt, "unit": "samples/s", "cores": int(blas), "kind": "port",
        "sample": "%d of the 1e6 UR10 samples, faithful reference structure (python per-sample loop around the C "
                  "regressor of oracle/, numpy scatter+permutation, np.diag(W.T@W), np.delete, 2x np.linalg.qr, "
                  "pinv); host has %d logical cpus, BLAS threads %d; stage seconds %s" % (
                      n_cpu, os.cpu_count(), blas, {k: round(x, 2) for k, x in stages.items()}),
    }


def cpu_baseline_fast(flat, seed, n_cpu):
    """The same pass with the host used well (OpenMP regressor in C, QR without Q): SURVEY 8(d) "fair-fast"."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import cpu_baseline as cb

    rng = np.random.default_rng(seed)
    q, v, a = (rng.uniform(-6, 6, (n_cpu, 6)) for _ in range(3))
    tau = rng.standard_normal(6 * n_cpu)
    cb.fast_pass(flat, q[:2000], v[:2000], a[:2000], tau[:12000])  # thread pools up
    t0 = time.perf_counter()
    _, stages = cb.fast_pass(flat, q, v, a, tau)
    dt = time.perf_counter() - t0
    return {"value": n_cpu / dt, "unit": "samples/s", "cores": os.cpu_count(), "kind": "port",
            "sample": "%d of the 1e6 UR10 samples, fair-fast structure (one OpenMP C call for the batch regressor, einsum "
                      "column norms, np.linalg.qr(mode='r') of [W_e tau], regrouped QR of the triangle, triangular solves); "
                      "stage seconds %s" % (n_cpu, {k: round(x, 2) for k, x in stages.items()})}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--samples", type=int, default=1_000_000, help="samples per GPU")
    ap.add_argument("--cpu-samples", type=int, default=300000)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--exchange", default="rccl", choices=["rccl", "torch"])
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node %d for --gpus %d" % (args.gpus, args.gpus))

    from figaroh_plus_amd import _lib
    from figaroh_plus_amd.dist import exchange_from_env
    from figaroh_plus_amd.pipeline import IdentificationPipeline
    from figaroh_plus_amd.tools.robot import Robot

    lib = _lib.load()
    if _lib.device_count() == 0:
        raise SystemExit("bench.py needs a HIP device: libfigh has no CPU path")
    _lib.check(lib.figh_device_set(local_rank % _lib.device_count()))
    exchange, xinfo = exchange_from_env(args.exchange)

    def barrier():
        _lib.synchronize()
        if world > 1:
            import torch.distributed as dist
            dist.barrier()

    with open(os.path.join(ROOT, "tests", "golden", "cfg2_ur10.json")) as f:
        meta = json.load(f)
    robot = Robot.from_flat("ur10")
    param = meta["param"]
    params_std = dict(zip(meta["names_std"], meta["phi_ref_raw"]))
    N = args.samples
    rng = np.random.default_rng(20250410 + 2 + 1000 * rank)
    q, v, a = (rng.uniform(-6, 6, (N, 6)) for _ in range(3))
    pipe = IdentificationPipeline(robot, param, params_std=params_std, exchange=exchange)
    pipe.set_samples(q, v, a)
    phi_ref = np.array([float(x) for x in meta["phi_ref_raw"]])
    d_tau = pipe.set_tau_from_parameters(phi_ref, noise_std=0.05, seed=rank)
    del q, v, a

    out = None
    # live HIP-event timing of the dominant kernels on the library stream (level 1: four event records per pass),
    # switched on during warm-up already so that the event pool exists before the timed region
    _lib.profile_enable(True, level=1)
    for _ in range(args.warmup):
        out = pipe.run()
    _lib.profile_reset()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = pipe.run()
    barrier()
    dt = time.perf_counter() - t0
    _lib.profile_enable(False)
    if world > 1:
        import torch
        import torch.distributed as dist
        t = torch.tensor([dt], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    # sanity: the step produced the reference's structural result
    ok = out["idx_base"] == meta_idx(meta) and out["params_base"] == meta["params_base"]
    n_kept = len(out["params_r"])
    kern = {}
    for name in ("regressor_chain", "tsqr"):
        cnt, ms = _lib.profile_get(name)
        if cnt:
            kern[name] = {"launches": cnt, "avg_ms": ms / cnt}
    # the small launches (merge levels, regrouped n x n factorisation, collectives) are timed in two extra,
    # untimed passes so that their event records do not sit in the timed region
    _lib.profile_enable(True, level=2)
    _lib.profile_reset()
    for _ in range(2):
        pipe.run()
    for name in ("tsqr_reduce", "tsqr_small", "gather_cols", "rccl_allgather", "rccl_allreduce"):
        cnt, ms = _lib.profile_get(name)
        if cnt:
            kern[name] = {"launches_per_step": cnt / 2.0, "avg_ms": ms / cnt, "timed_region": False}
    _lib.profile_enable(False)
    rows_per_sample, ncols = 6, 84
    bytes_per_sample = 8 * 18 + 8 * rows_per_sample * ncols                      # 4176 B
    flops_per_sample = 2 * rows_per_sample * (n_kept + 1) ** 2                   # 2 m n^2
    roof = {}
    if "regressor_chain" in kern:
        sec = kern["regressor_chain"]["avg_ms"] * 1e-3
        roof["regressor_chain"] = {"bound": "hbm", "achieved": bytes_per_sample * N / sec / 1e9, "peak": HBM_PEAK_GBS,
                                   "unit": "GB/s", "traffic": None}
    if "tsqr" in kern:
        sec = kern["tsqr"]["avg_ms"] * 1e-3
        roof["tsqr"] = {"bound": "mfma", "achieved": flops_per_sample * N / sec / 1e12, "peak": FP64_PEAK_TFLOPS,
                        "unit": "TFLOP/s", "traffic": None}
    # HBM bytes per launch from the committed rocprofv3 PMC passes of this same command (FETCH_SIZE x2 as the
    # gfx950 correction + WRITE_SIZE, tools/pmc_summary.py); PMC cannot be read from inside the process
    try:
        with open(os.path.join(ROOT, "profiles", "r01_pmc_summary.json")) as f:
            pmc = json.load(f)
        if N == 1_000_000:
            for key, kname in (("regressor_chain", "regressor_chain_kernel<6, false, true>"),
                               ("tsqr", "tsqr2_kernel<4, 4, false, false, true>")):
                if key in roof and kname in pmc and "hbm_bytes" in pmc[kname]:
                    roof[key]["traffic"] = pmc[kname]["hbm_bytes"]
                    roof[key]["traffic_source"] = "profiles/r01_pmc_summary.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE)"
    except Exception:
        pass
    if "tsqr" in roof:
        roof["tsqr"]["pipe"] = ("fp64 VALU (v_fmac_f64 with a DPP row_newbcast operand + permlane swaps); v_mfma_f64_16x16x4 has the "
                              "same 78.6 TFLOP/s peak and does not overlap with fp64 VALU work (tools/microbench/latency.hip)")
        roof["tsqr"]["algorithmic_flops_per_sample"] = flops_per_sample
    if "regressor_chain" in roof:
        roof["regressor_chain"]["algorithmic_bytes_per_sample"] = bytes_per_sample
    for r in roof.values():
        r["frac"] = r["achieved"] / r["peak"]
    main = {k: v for k, v in kern.items() if "launches" in v}
    dominant = max(main, key=lambda k: main[k]["avg_ms"] * main[k]["launches"]) if main else None
    if rank == 0:
        line = {
            "metric": "samples/sec regressor build + TSQR solve, UR10 6-DoF",
            "value": N * world * args.steps / dt,
            "unit": "samples/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * dt / args.steps,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": "BASELINE configs[1]: UR10 6-DoF, %d synthetic (q,qd,qdd) samples per GPU, full inertial "
                            "regressor materialised (6N x 84) + elimination + Householder TSQR base params + LS" % N,
                "samples_per_gpu": N, "columns": ncols, "kept_columns": n_kept, "base_parameters": len(out["idx_base"]),
                "collective": xinfo["collective"], "device": _lib.device_info()["name"],
                "result_matches_reference": bool(ok),
            },
            "roofline": dict(roof.get(dominant, {}), kernel=dominant) if dominant in roof else None,
            "kernels": {k: dict(kern[k], **roof.get(k, {})) for k in kern},
        }
        if not args.no_cpu_baseline and world == 1:
            line["cpu_baseline"] = cpu_baseline(robot.model.to_flat(), 7, args.cpu_samples)
            line["cpu_baseline_fast"] = cpu_baseline_fast(robot.model.to_flat(), 7, args.cpu_samples)
        print(json.dumps(line))
    if world > 1:
        barrier()
        if hasattr(exchange, "close"):
            exchange.close()


def meta_idx(meta):
    names = meta["params_r"]
    base_first = [p.split(" ")[0] for p in meta["params_base"]]
    return [names.index(b) for b in base_first]


if __name__ == "__main__":
    main()
