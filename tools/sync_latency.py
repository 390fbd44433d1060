"""Host-observed time of launch + wait against the kernel's own duration (HIP events) for the TALOS regressor pass
(25 ms) and the TSQR level 0 (180 ms): where do the 10-60 ms per step go that the kernels do not account for?"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from figaroh_plus_amd import _lib
from figaroh_plus_amd.device import GpuMatrix
from figaroh_plus_amd.tools.randomdata import sample_inputs
from figaroh_plus_amd.tools.regressor import regressor_flags
from figaroh_plus_amd.tools.robot import Robot

meta = json.load(open(os.path.join(ROOT, "tests", "golden", "cfg4_talos.json")))
robot = Robot.from_flat("talos")
N = int(float(sys.argv[1])) if len(sys.argv) > 1 else 4_000_000
q, v, a = sample_inputs(robot.model, N, np.random.default_rng(1), 1.5, 2, 5)
d_q, d_v, d_a = (_lib.DeviceArray.from_host(x.reshape(-1)) for x in (q, v, a))
mode, flags, ft = regressor_flags(meta["param"], False)
h = robot.device_model()
rps, ncols = h.shape(mode, flags)
W = GpuMatrix.empty(rps * N, 16 * (robot.model.njoints - 1))
d_colsq = _lib.DeviceArray((ncols,), np.float64)
_lib.profile_enable(True, level=2)
for it in range(8):
    _lib.profile_reset()
    t0 = time.perf_counter()
    _lib.regressor_build_padded(h, mode, flags, ft, N, d_q, d_v, d_a, W.buf, W.ld, d_colsq)
    t1 = time.perf_counter()
    if it % 2 == 0:
        _lib.synchronize()
    else:
        d_colsq.to_host()
    t2 = time.perf_counter()
    cnt, ms = _lib.profile_get("regressor_tree")
    print("iter %d (%s): launch call %.3f ms, wait %.3f ms, kernel by events %.3f ms" % (
        it, "synchronize" if it % 2 == 0 else "D2H copy", 1e3 * (t1 - t0), 1e3 * (t2 - t1), ms / max(cnt, 1)), flush=True)
    time.sleep(0.05 * (it % 3))
