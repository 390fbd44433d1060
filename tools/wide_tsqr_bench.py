"""Level-0 TSQR on caller-supplied tall matrices at the wide column counts (human 191, TIAGo 241, TALOS 331, SIP 400):
correctness against LAPACK on a slice, kernel time by HIP events, algorithmic TFLOP/s (2 m n^2) against 78.6.
usage: python tools/wide_tsqr_bench.py [rows] [n ...]      (FIGH_LIB_PATH selects another build of the library)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from figaroh_plus_amd import _lib

rows = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1_000_000
ns = [int(x) for x in sys.argv[2:]] or [191, 241, 331, 400]
rng = np.random.default_rng(3)
print("device", _lib.device_info()["name"], "lib", os.path.basename(_lib.LIB_PATH), "cfg", os.environ.get("FIGH_WY_CFG"), flush=True)
for n in ns:
    t0 = time.perf_counter()
    small = rng.standard_normal((20000, n))
    # the big matrix is the small one tiled with per-tile column scalings (cheap to generate, full rank)
    reps = (rows + 19999) // 20000
    A = np.empty((reps * 20000, n))
    for r in range(reps):
        A[r * 20000:(r + 1) * 20000] = small * (1.0 + 0.01 * r)
    A = A[:rows]
    d_A = _lib.DeviceArray.from_host(A.reshape(-1))
    d_R = _lib.DeviceArray((n * n,), np.float64)
    # correctness on the first 20000 rows
    _lib.tsqr(d_A, 20000, n, None, n, None, None, d_R)
    R = d_R.to_host().reshape(n, n)
    Rref = np.linalg.qr(small, mode="r")
    scale = np.abs(Rref).max()
    e_diag = np.abs(np.abs(np.diag(R)) - np.abs(np.diag(Rref))).max() / scale
    e_gram = np.abs(R.T @ R - small.T @ small).max() / np.abs(small.T @ small).max()
    low = np.abs(np.tril(R, -1)).max()
    for _ in range(2):
        _lib.tsqr(d_A, rows, n, None, n, None, None, d_R)
    _lib.synchronize()
    _lib.profile_enable(True); _lib.profile_reset()
    K = 5
    t1 = time.perf_counter()
    for _ in range(K):
        _lib.tsqr(d_A, rows, n, None, n, None, None, d_R)
    _lib.synchronize()
    wall = (time.perf_counter() - t1) / K
    c0, ms0 = _lib.profile_get("tsqr"); c1, ms1 = _lib.profile_get("tsqr_reduce")
    _lib.profile_enable(False)
    lvl0 = ms0 / max(c0, 1)
    fl = 2.0 * rows * n * n
    print("n=%d rows=%d: |diag| err %.1e gram err %.1e lower %.1e | level0 %.2f ms = %.1f TF/s (%.1f%% of 78.6) | merges %.2f ms (%d launches) | wall %.2f ms  [gen %.0fs]" % (
        n, rows, e_diag, e_gram, low, lvl0, fl / lvl0 / 1e9, 100 * fl / lvl0 / 1e9 / 78.6, ms1 / K, c1 // K, wall * 1e3, t1 - t0), flush=True)
    del d_A, A
