"""65 .. 80 columns: the register-tile kernel's 48-row form (tsqr2_kernel<5, 3>) against the blocked kernel (tsqr_wy_kernel) on the
same tall matrix.  The library routes nc <= 80 to the former; the blocked kernel is reached here by handing it the same columns
plus zero columns up to 81 (null pivots: a norm per tile each).  Round 6 question: would the human model's force rows (3e7 x 77,
19 ms) and TIAGo's mid row blocks be better off with the blocked kernel?   python tools/wide_vs_narrow_5chunks.py [rows]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from figaroh_plus_amd import _lib

rows = int(float(sys.argv[1])) if len(sys.argv) > 1 else 30_000_000
ld = 96
rng = np.random.default_rng(0)
blk = rng.standard_normal((1 << 20, ld))
blk[:, 77:] = 0.0
d_W = _lib.DeviceArray((rows * ld,), np.float64)
lib = _lib.load()
for lo in range(0, rows, 1 << 20):  # tile the block over the matrix with a row-dependent scale (not rank-deficient across tiles)
    n = min(1 << 20, rows - lo)
    part = blk[:n] * (1.0 + 1e-3 * (lo >> 20))
    _lib.check(lib.figh_memcpy_h2d(d_W.ptr + lo * ld * 8, part.ctypes.data, part.nbytes))
_lib.null_pivots(1e-8).__enter__()
for n in (64, 77, 80, 81, 96):
    d_R = _lib.DeviceArray((n * n,), np.float64)
    for rep in range(3):
        _lib.synchronize()
        t0 = time.perf_counter()
        _lib.tsqr(d_W, rows, ld, None, n, None, None, d_R)
        _lib.synchronize()
        dt = time.perf_counter() - t0
    live = min(n, 77)
    print("n = %2d (%d live columns)  %8.3f ms   %.1f TFLOP/s on 2 m %d^2" % (n, live, 1e3 * dt, 2.0 * rows * live * live / dt / 1e12, live))
