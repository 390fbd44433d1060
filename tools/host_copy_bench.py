"""Rates of figh_memcpy_h2d / figh_memcpy_d2h between NumPy arrays and HBM (the drop-in boundary), per destination kind.
python tools/host_copy_bench.py [GB]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from figaroh_plus_amd import _lib
from figaroh_plus_amd.device import host_empty

gb = float(sys.argv[1]) if len(sys.argv) > 1 else 4.032
n = int(gb * 1e9 / 8)
lib = _lib.load()
d = _lib.DeviceArray((n,), np.float64)
_lib.check(lib.figh_memset(d.ptr, 0, d.nbytes))
_lib.synchronize()


def d2h(make, label, reps=2):
    for r in range(reps):
        t0 = time.perf_counter()
        out = make()
        t1 = time.perf_counter()
        _lib.check(lib.figh_memcpy_d2h(out.ctypes.data, d.ptr, 8 * n))
        t2 = time.perf_counter()
        print("D2H %-44s alloc %7.1f ms copy %7.1f ms  %5.1f GB/s (with alloc %5.1f)" % (
            label + (" (again, same kind)" if r else ""), 1e3 * (t1 - t0), 1e3 * (t2 - t1), 8 * n / (t2 - t1) / 1e9, 8 * n / (t2 - t0) / 1e9))
        del out


d2h(lambda: np.empty(n), "fresh np.empty")
d2h(lambda: host_empty(n), "fresh host_empty (huge pages)")
touched = np.zeros(n)
d2h(lambda: touched, "touched ndarray")
pin = _lib.PinnedArray(n)
d2h(lambda: pin.array, "page-locked (figh_host_alloc)")
for m in (6_000_000, 34_000_000, n):
    src = np.random.default_rng(0).standard_normal(m)
    dd = _lib.DeviceArray((m,), np.float64)
    for r in range(2):
        _lib.synchronize()
        t0 = time.perf_counter()
        _lib.check(lib.figh_memcpy_h2d(dd.ptr, src.ctypes.data, src.nbytes))
        _lib.synchronize()
        dt = time.perf_counter() - t0
        print("H2D %6.0f MB from a touched ndarray   %7.1f ms  %5.1f GB/s" % (src.nbytes / 1e6, 1e3 * dt, src.nbytes / dt / 1e9))
    dd.free()
