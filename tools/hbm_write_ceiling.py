import sys, time
sys.path.insert(0, '/root/repo')
import numpy as np
from figaroh_plus_amd import _lib
lib = _lib.load()
n = 6_000_000 * 84
d = _lib.DeviceArray((n,))
for _ in range(3):
    _lib.check(lib.figh_memset(d.ptr, 0, n * 8)); _lib.synchronize()
ts = []
for _ in range(10):
    t0 = time.perf_counter(); _lib.check(lib.figh_memset(d.ptr, 0, n * 8)); _lib.synchronize(); ts.append(time.perf_counter() - t0)
print("memset 4.03 GB: best %.3f ms -> %.2f TB/s" % (min(ts) * 1e3, n * 8 / min(ts) / 1e12))
e = _lib.DeviceArray((n,))
for _ in range(3):
    _lib.check(lib.figh_memcpy_d2d(e.ptr, d.ptr, n * 8)); _lib.synchronize()
ts = []
for _ in range(10):
    t0 = time.perf_counter(); _lib.check(lib.figh_memcpy_d2d(e.ptr, d.ptr, n * 8)); _lib.synchronize(); ts.append(time.perf_counter() - t0)
print("copy 4.03 GB: best %.3f ms -> %.2f TB/s (read+write)" % (min(ts) * 1e3, 2 * n * 8 / min(ts) / 1e12))
