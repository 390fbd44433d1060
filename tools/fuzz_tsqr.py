#!/usr/bin/env python3
"""Randomised differential run of figh_tsqr against the Gram matrix / LAPACK: random shapes, rank deficiency, tau, row weights,
gathered columns, null-pivot rule on and off.  usage: python tools/fuzz_tsqr.py [cases] [seed]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from figaroh_plus_amd import _lib  # noqa: E402
from figaroh_plus_amd.tools.qrdecomposition import rfactor  # noqa: E402

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
bad = 0
for k in range(cases):
    n = int(rng.choice([rng.integers(1, 17), rng.integers(17, 65), rng.integers(65, 81), rng.integers(81, 200), rng.integers(200, 420)]))
    rows = int(rng.choice([rng.integers(1, 64), rng.integers(64, 2000), rng.integers(2000, 60000), rng.integers(60000, 400000)]))
    if n > 200 and rows > 100000:
        rows = 100000
    ld = n + int(rng.integers(0, 20))
    W = rng.standard_normal((rows, ld)) * rng.uniform(0.1, 30.0, ld)
    cols = np.sort(rng.choice(ld, n, replace=False)).astype(np.int32) if ld > n else None
    ndep = int(rng.integers(0, max(1, n // 3))) if n > 3 and rng.random() < 0.6 else 0
    A = W if cols is None else W[:, cols]
    dep = []
    if ndep:
        dep = np.sort(rng.choice(np.arange(1, n), ndep, replace=False))
        for j in dep:
            src = [c for c in range(j) if c not in set(dep.tolist())]
            pick = rng.choice(src, min(3, len(src)), replace=False)
            A[:, j] = A[:, pick] @ rng.uniform(-2, 2, len(pick))
        if cols is not None:
            W[:, cols] = A
    t = rng.standard_normal(rows) if rng.random() < 0.6 else None
    wts = None
    if rows % 4 == 0 and rng.random() < 0.3:
        wts = rng.uniform(0.5, 2.0, 4)
    tol = float(rng.choice([0.0, 1e-8 / 64]))
    _lib.tsqr_null_pivot_tol(tol)
    try:
        R = rfactor(W, tau=t, col_idx=cols, block_weight=wts)
    finally:
        _lib.tsqr_null_pivot_tol(0.0)
    M = A if t is None else np.c_[A, t]
    if wts is not None:
        M = M * np.repeat(wts, rows // 4)[:, None]
    G = M.T @ M
    err = np.abs(R.T @ R - G).max() / max(np.abs(G).max(), 1e-300)
    ok = np.array_equal(R, np.triu(R)) and np.isfinite(R).all() and err <= 2e-12
    if ok and rows >= M.shape[1] and ndep == 0:
        ref = np.linalg.qr(M, mode="r")
        ok = np.abs(np.abs(np.diag(R)) - np.abs(np.diag(ref))).max() <= 1e-8 * np.abs(np.diag(ref)).max()
    if ok and ndep and rows > 4 * n:
        d = np.abs(np.diag(R))[:n]
        ok = d[dep].max() <= 1e-8 and (np.delete(d, dep) > 1e-8).all()
    if not ok:
        bad += 1
        print("FAIL case %d: n %d rows %d ld %d ndep %d tau %s weights %s tol %.1e err %.2e" % (k, n, rows, ld, ndep, t is not None, wts is not None, tol, err))
print("%d cases, %d failures" % (cases, bad))
