#!/usr/bin/env python3
"""Level 0 of the register-tile TSQR for 65 .. 80 columns: 48-row tiles + register chunk (two waves per SIMD) against the
64-row form (fewer rows than eight waves per CU need: one wave per SIMD), on the human model's force-row shape and TIAGo's widest narrow blocks."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from figaroh_plus_amd import _lib  # noqa: E402
from figaroh_plus_amd.tools.qrdecomposition import rfactor  # noqa: E402
from figaroh_plus_amd.device import GpuMatrix, to_device  # noqa: E402

for n, rows, ld in ((76, 3000000, 96), (79, 1000000, 192), (65, 1000000, 128), (80, 500000, 80)):
    rng = np.random.default_rng(n)
    A = rng.standard_normal((rows, ld))
    cols = np.sort(rng.choice(ld, n, replace=False)).astype(np.int32) if ld > n else None
    t = rng.standard_normal(rows) if n < 80 else None
    Wd, _ = to_device(A)
    out = {}
    for mode in ("T54", "T53", "T54b", "T53b"):  # (ablation build: FIGH_LIB_PATH=figaroh_plus_amd/libfigh_ab.so)
        os.environ["FIGH_T53"] = "1" if mode.startswith("T53") else "0"
        R = rfactor(Wd, tau=t, col_idx=cols)
        _lib.profile_enable(True, 1)
        _lib.profile_reset()
        for _ in range(5):
            R = rfactor(Wd, tau=t, col_idx=cols)
        cnt, ms = _lib.profile_get("tsqr")
        _lib.profile_enable(False)
        out[mode] = (R, ms / max(cnt, 1))
    os.environ.pop("FIGH_T53", None)
    M = A if cols is None else A[:, cols]
    if t is not None:
        M = np.c_[M, t]
    G = M[:200000].T @ M[:200000] if rows > 200000 else M.T @ M
    R3, R4 = out["T53"][0], out["T54"][0]
    d = np.abs(np.abs(np.diag(R3)) - np.abs(np.diag(R4))).max() / np.abs(np.diag(R4)).max()
    g = np.abs(R3.T @ R3 - R4.T @ R4).max() / np.abs(R4.T @ R4).max()
    print("n %d rows %d ld %d: level 0 %.3f / %.3f ms (48-row, register chunk) vs %.3f / %.3f ms (64-row); |diag| rel diff %.1e, RtR rel diff %.1e" % (
        n, rows, ld, out["T53"][1], out["T53b"][1], out["T54"][1], out["T54b"][1], d, g))
