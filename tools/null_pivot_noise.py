#!/usr/bin/env python3
"""How large do the pivots of exactly dependent columns get, with and without the null-pivot rule?"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from figaroh_plus_amd import _lib  # noqa: E402
from figaroh_plus_amd.tools.qrdecomposition import rfactor  # noqa: E402


def _rank_deficient(rng, rows, n, ndep):
    A = rng.standard_normal((rows, n)) * rng.uniform(0.5, 20.0, n)
    dep = np.sort(rng.choice(np.arange(1, n), ndep, replace=False))
    for j in dep:
        src = [k for k in range(j) if k not in set(dep.tolist())]
        pick = rng.choice(src, min(3, len(src)), replace=False)
        A[:, j] = A[:, pick] @ rng.uniform(-2.0, 2.0, len(pick))
    return A, dep


for n, ndep, rows in ((50, 13, 64), (50, 13, 6400), (50, 13, 640000), (331, 96, 64), (331, 96, 64 * 512), (331, 96, 64 * 512 * 8),
                      (331, 96, 64 * 512 * 64), (331, 0, 64 * 512 * 64), (191, 27, 64 * 512 * 64)):
    rng = np.random.default_rng(n + rows)
    A, dep = _rank_deficient(rng, rows, n, ndep) if ndep else (rng.standard_normal((rows, n)), np.array([0]))
    res = []
    for tol in (0.0, 1e-8 / 64, 1e-8 / 4):
        _lib.tsqr_null_pivot_tol(tol)
        R = rfactor(A)
        _lib.profile_enable(True, 1)
        _lib.profile_reset()
        R = rfactor(A)
        cnt, ms = _lib.profile_get("tsqr")
        _lib.profile_enable(False)
        d = np.abs(np.diag(R))
        res.append("tol %.2e: dep max %.2e median %.2e level-0 %.3f ms" % (tol, d[dep].max(), np.median(d[dep]), ms / max(cnt, 1)))
    _lib.tsqr_null_pivot_tol(0.0)
    print(n, rows, " | ".join(res))
