#!/usr/bin/env python3
"""Can K1 (HBM-write-bound) and the level-0 TSQR (fp64-VALU-bound) share the chip?  Ablation build only
(FIGH_LIB_PATH=.../libfigh_ab.so): W1 is built, then the TSQR of W1 (stream 1) and K1 into a second buffer W2 (stream 0)
are put in flight together, in both launch orders, and the wall time is compared with the two alone."""
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from figaroh_plus_amd import _lib  # noqa: E402
from figaroh_plus_amd.tools.regressor import regressor_flags  # noqa: E402
from figaroh_plus_amd.tools.robot import Robot  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
meta = json.load(open(os.path.join(ROOT, "tests", "golden", "cfg2_ur10.json")))
robot = Robot.from_flat("ur10")
lib = _lib.load()
lib.figh_ab_stream_select.argtypes = [C.c_int]
rng = np.random.default_rng(0)
q, v, a = (rng.uniform(-6, 6, (N, 6)) for _ in range(3))
d_q, d_v, d_a = (_lib.DeviceArray.from_host(x.reshape(-1)) for x in (q, v, a))
mode, flags, ft_mask = regressor_flags(meta["param"], False)
dm = robot.device_model()
rps, ncols = dm.shape(mode, flags)
W1 = _lib.DeviceArray((rps * N * ncols,), np.float64)
W2 = _lib.DeviceArray((rps * N * ncols,), np.float64)
cs1, cs2 = _lib.DeviceArray((ncols,), np.float64), _lib.DeviceArray((ncols,), np.float64)
d_sel = _lib.DeviceArray((2 + 2 * ncols,), np.int32)
cap = ncols + 1
d_rows = _lib.DeviceArray(((cap + 1) * cap,), np.float64)


def k1(W, cs):
    _lib.regressor_build(dm, mode, flags, ft_mask, N, d_q, d_v, d_a, W, ncols, cs)


def tsqr():
    _lib.tsqr_selected(W1, rps * N, ncols, cs1, ncols, 1e-6, 14, rps, 49, None, 1e-8, d_sel, d_rows)


def sync():
    _lib.check(lib.figh_ab_device_sync())


def timed(fn, reps=7):
    ts = []
    for _ in range(reps):
        sync()
        t0 = time.perf_counter()
        fn()
        sync()
        ts.append(time.perf_counter() - t0)
    return 1e3 * float(np.median(ts[2:]))


k1(W1, cs1); k1(W2, cs2); tsqr(); sync()


def both(order):
    def run():
        for what in order:
            if what == "t":
                lib.figh_ab_stream_select(1); tsqr()
            else:
                lib.figh_ab_stream_select(0); k1(W2, cs2)
        lib.figh_ab_stream_select(0)
    return run


res = {"N": N, "k1_alone_ms": timed(lambda: k1(W2, cs2)), "tsqr_alone_ms": timed(tsqr),
       "tsqr_then_k1_ms": timed(both("tk")), "k1_then_tsqr_ms": timed(both("kt")),
       "serial_ms": timed(lambda: (k1(W2, cs2), tsqr()))}
print(json.dumps(res))
