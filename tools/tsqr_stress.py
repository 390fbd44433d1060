"""Randomised stress of figh_tsqr on the blocked-kernel range: random column counts 81..512 (every geometry, LDS-chunk
forms included), random row counts incl. ragged and fewer-rows-than-columns, column gathers, tau column, row-block
weights, blocks of structurally zero leading columns and exactly dependent columns; checked against NumPy Gram
matrices / LAPACK pivots.   usage: python tools/tsqr_stress.py [cases] [seed]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from figaroh_plus_amd.tools.qrdecomposition import rfactor

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 7)
worst = 0.0
for k in range(cases):
    n = int(rng.choice([rng.integers(81, 513), rng.choice([192, 193, 256, 257, 320, 321, 336, 337, 384, 385, 400, 401])]))
    rows = int(rng.choice([rng.integers(1, 400), rng.integers(400, 30000), 64 * rng.integers(1, 300), 48 * rng.integers(1, 300)]))
    ncols_src = n + int(rng.integers(0, 40))
    A = rng.standard_normal((rows, ncols_src)) * rng.uniform(0.1, 30.0, ncols_src)
    mode = k % 4
    if mode == 1 and rows > 200:       # structurally zero leading columns for the second half of the rows
        A[rows // 2:, :n // 3] = 0.0
    cols = np.sort(rng.choice(ncols_src, n, replace=False)) if ncols_src > n else None
    src = cols if cols is not None else np.arange(n)
    if mode == 2:                      # exactly dependent columns (among the gathered ones)
        A[:, src[5]] = A[:, src[3]] + 2.0 * A[:, src[4]]
        A[:, src[n - 1]] = A[:, src[0]]
    As = A[:, cols] if cols is not None else A
    use_tau = bool(rng.integers(0, 2))
    t = rng.standard_normal(rows) if use_tau else None
    nblk = int(rng.choice([1, 2, 4])) if rows % 4 == 0 else 1
    w = rng.uniform(0.5, 2.0, nblk)
    R = rfactor(A, tau=t, col_idx=cols, block_weight=w if nblk > 1 or k % 3 == 0 else None)
    scale = np.repeat(w, rows // nblk) if (nblk > 1 or k % 3 == 0) else np.ones(rows)
    M = (np.c_[As, t] if use_tau else As) * scale[:, None]
    G = M.T @ M
    err = np.abs(R.T @ R - G).max() / np.abs(G).max()
    low = np.abs(np.tril(R, -1)).max()
    worst = max(worst, err)
    ok = err <= 2e-12 and low == 0.0
    if mode == 2 and rows > 2 * n:     # dependent pivots must come out tiny (the rank decision of the path)
        d = np.abs(np.diag(R))[:n]
        ok = ok and d[5] <= 1e-9 * np.abs(G).max() ** 0.5 and d[n - 1] <= 1e-9 * np.abs(G).max() ** 0.5
    print("case %2d n=%3d rows=%5d gather=%s tau=%s blocks=%d mode=%d: gram err %.1e %s" % (
        k, n, rows, cols is not None, use_tau, nblk, mode, err, "ok" if ok else "FAIL"), flush=True)
    if not ok:
        sys.exit(1)
print("all %d cases ok, worst relative Gram error %.1e" % (cases, worst))
