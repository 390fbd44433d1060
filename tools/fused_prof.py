#!/usr/bin/env python3
"""In-kernel time buckets of the fused K1 + TSQR launch (ablation build only: FIGH_LIB_PATH=figaroh_plus_amd/libfigh_ab.so).

  FIGH_LIB_PATH=figaroh_plus_amd/libfigh_ab.so python tools/fused_prof.py [N] [FIGH_FUSED_OPTS value ...]
"""
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from figaroh_plus_amd import _lib  # noqa: E402
from figaroh_plus_amd.pipeline import IdentificationPipeline  # noqa: E402
from figaroh_plus_amd.tools.robot import Robot  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
opts = [int(x) for x in sys.argv[2:]] or [0]
with open(os.path.join(ROOT, "tests", "golden", "cfg2_ur10.json")) as f:
    meta = json.load(f)
robot = Robot.from_flat("ur10")
rng = np.random.default_rng(1)
q, v, a = (rng.uniform(-6, 6, (N, 6)) for _ in range(3))
params_std = dict(zip(meta["names_std"], meta["phi_ref_raw"]))
phi_ref = np.array([float(x) for x in meta["phi_ref_raw"]])
lib = _lib.load()
prof = getattr(lib, "figh_fused_prof", None)
if prof is not None:
    prof.restype = C.c_int
    prof.argtypes = [C.POINTER(C.c_double), C.c_int]
names = {0: "P wait buffer", 1: "P inputs+forward", 2: "P emit row", 3: "P stream-out", 4: "P col norms", 5: "P total",
         8: "C wait tile", 9: "C gather", 10: "C column steps", 11: "C total", 12: "C tiles"}
for o in opts:
    os.environ["FIGH_FUSED_OPTS"] = str(o)
    pipe = IdentificationPipeline(robot, meta["param"], params_std=params_std, fuse=True)
    pipe.set_samples(q, v, a)
    pipe.set_tau_from_parameters(phi_ref, noise_std=0.05, seed=0)
    for _ in range(3):
        pipe.run()
    buf = (C.c_double * 16)()
    if prof is not None:
        prof(buf, 1)
    _lib.profile_enable(True, level=1)
    _lib.profile_reset()
    reps = 10
    _lib.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        pipe.run()
    _lib.synchronize()
    dt = (time.perf_counter() - t0) / reps
    cnt, ms = _lib.profile_get("fused_chain_tsqr")
    _lib.profile_enable(False)
    print("opts %d: step %.3f ms, fused kernel %.3f ms (%d launches, fused passes %d)" % (
        o, 1e3 * dt, ms / max(cnt, 1), cnt, pipe.fused_passes))
    if prof is not None:
        prof(buf, 1)
        vals = list(buf)
        for base, tot in ((0, 5), (8, 11)):
            if vals[tot] > 0:
                print("   " + ", ".join("%s %.1f%%" % (names[k], 100.0 * vals[k] / vals[tot])
                                        for k in range(base, tot) if k in names))
        if vals[12] > 0:
            print("   LDS read round trip (one dependent ds_read_b64 + wait, incl. 2 counter reads): producer %.0f, consumer %.0f "
                  "ticks" % (vals[6] / vals[12], vals[13] / vals[12]))
            print("   ticks per tile: producer %.0f (x1 wave), consumer busy %.0f; tiles %d" % (
                vals[5] / vals[12] * reps * 0 + vals[5] / (vals[12]), (vals[9] + vals[10]) / vals[12], vals[12] / reps))
    del pipe
