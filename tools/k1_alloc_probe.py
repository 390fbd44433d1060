"""Does the K1 time depend on where W lands?  Re-allocate W several times in one process and time the kernel."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from figaroh_plus_amd import _lib
from figaroh_plus_amd.tools.robot import Robot
from figaroh_plus_amd.tools.regressor import regressor_flags
meta = json.load(open(ROOT + '/tests/golden/cfg2_ur10.json'))
robot = Robot.from_flat('ur10'); N = 1000000
rng = np.random.default_rng(1); q, v, a = (rng.uniform(-6, 6, (N, 6)) for _ in range(3))
mode, flags, ft = regressor_flags(meta['param'], False); dm = robot.device_model()
dq, dv, da = (_lib.DeviceArray.from_host(x.reshape(-1)) for x in (q, v, a))
dc = _lib.DeviceArray((84,))
junk = []
for trial in range(int(os.environ.get("FIGH_PROBE_TRIALS", "12"))):
    W = _lib.DeviceArray((6 * N * 84,))
    for _ in range(3): _lib.regressor_build(dm, mode, flags, ft, N, dq, dv, da, W, 84, dc)
    _lib.synchronize(); t0 = time.perf_counter()
    for _ in range(10): _lib.regressor_build(dm, mode, flags, ft, N, dq, dv, da, W, 84, dc)
    _lib.synchronize(); dt = (time.perf_counter() - t0) / 10
    lib = _lib.load()
    for _ in range(2): _lib.check(lib.figh_memset(W.ptr, 0, W.nbytes))
    _lib.synchronize(); t0 = time.perf_counter()
    for _ in range(10): _lib.check(lib.figh_memset(W.ptr, 0, W.nbytes))
    _lib.synchronize(); dm_ = (time.perf_counter() - t0) / 10
    print("trial %d  W at 0x%x (mod 2MB %d, mod 1GB %d MB)  K1 %.3f ms  memset %.3f ms" % (trial, W.ptr, W.ptr % (2 << 20), (W.ptr % (1 << 30)) >> 20, dt * 1e3, dm_ * 1e3), flush=True)
    if trial % 2 == 0:
        junk.append(_lib.DeviceArray((1 << 20) * (trial + 3),))  # perturb the next placement
    W.free()
