#!/usr/bin/env python3
"""Strong-scaling ceiling from one GPU: the per-rank shard of an 8-GPU run is timed on the one GPU a
test box has -- bench.py with the config's sample count divided by 1, 2, 4, 8 -- and the speed-up the design can reach is

    predicted_speedup_P = t(N) / (t(N / P) + exchange_P)

exchange_P = what a rank does in addition when P > 1: all-reduce of the column norms, all-gather of the nc x nc triangle
(two latency-bound RCCL calls, ASSUMED 30 us each over xGMI -- they cannot be measured on one GPU) and the rank decision
on the stack of P triangles instead of one (measured here: figh_tsqr_merge_base of P triangles minus that of one).
No multi-GPU curve has been measured; this bounds it.   usage: python tools/shard_sweep.py [cfg2 cfg4 cfg5] > out.json"""
import json
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

RCCL_CALL_US = 30.0


def step_ms(cfg, samples, steps):
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--config", cfg, "--samples", str(samples), "--steps", str(steps),
           "--warmup", "2", "--no-cpu-baseline", "--scaling", "strong", "--strong-config="]
    out = subprocess.run(cmd, capture_output=True, text=True, check=True).stdout
    line = json.loads([l for l in out.splitlines() if l.startswith("{")][-1])
    return line["ms_per_step_median"], {k: v.get("avg_ms") for k, v in line["kernels"].items()}, line["config"]["kept_columns"]


def merge_extra_us(nc, P):
    """rank decision on a stack of P triangles minus on one (library event timing)."""
    from figaroh_plus_amd import _lib as lib
    rng = np.random.default_rng(0)
    res = {}
    for count in (1, P):
        stack = np.triu(rng.standard_normal((count, nc, nc)))
        d_stack = lib.DeviceArray.from_host(stack.reshape(-1))
        d_out = lib.DeviceArray(((nc + 1) * nc,))
        for _ in range(3):
            lib.tsqr_merge_base(d_stack, count, nc, nc - 1, 1e-8, d_out)
        import time
        lib.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            lib.tsqr_merge_base(d_stack, count, nc, nc - 1, 1e-8, d_out)
        lib.synchronize()
        res[count] = (time.perf_counter() - t0) / 20 * 1e6
    return max(0.0, res[P] - res[1]), res


def main():
    cfgs = sys.argv[1:] or ["cfg2", "cfg4", "cfg5"]
    report = {"rccl_call_us_assumed": RCCL_CALL_US, "configs": {}}
    for cfg in cfgs:
        n_total = bench.CONFIGS[cfg][2]
        steps = 10 if cfg == "cfg2" else 3
        rows = {}
        for P in (1, 2, 4, 8):
            ms, kern, kept = step_ms(cfg, n_total // P, steps)
            rows[P] = {"samples_per_rank": n_total // P, "ms_per_step": ms, "kernels_ms": kern}
        nc = kept + 1
        entry = {"samples_total": n_total, "nc": nc, "per_rank": rows, "predicted": {}}
        for P in (2, 4, 8):
            extra_us, raw = merge_extra_us(nc, P)
            ex_ms = (2 * RCCL_CALL_US + extra_us) / 1e3
            entry["predicted"][P] = {"exchange_ms": ex_ms, "merge_of_P_minus_one_us": extra_us,
                                     "speedup": rows[1]["ms_per_step"] / (rows[P]["ms_per_step"] + ex_ms)}
        report["configs"][cfg] = entry
        sys.stderr.write("%s: %s\n" % (cfg, {P: round(entry["predicted"][P]["speedup"], 2) for P in (2, 4, 8)}))
    print(json.dumps(report, indent=1))


if __name__ == "__main__":
    main()
