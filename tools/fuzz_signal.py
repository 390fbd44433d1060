#!/usr/bin/env python3
"""Randomised run of the device filters (figh_signal.hip) against SciPy: zero-phase decimation (scipy.signal.decimate(zero_phase=True)
per column and row block, what examples/staubli_TX40/identification.py:191-204 and examples/tiago/identification.py:142-187 do to every
column of W) and the Butterworth filtfilt of identification_tools.py:398-424, on random lengths / column counts / block counts / factors.
usage: python tools/fuzz_signal.py [cases] [seed]"""
import os
import sys
import traceback

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
from scipy import signal  # noqa: E402
from figaroh_plus_amd import _lib  # noqa: E402
from figaroh_plus_amd.identification.identification_tools import (_decimate_design, _filtfilt_device, decimate_joint_blocks,  # noqa: E402
                                                                   low_pass_filter_data)

_lib.load()
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
bad = 0
for k in range(cases):
    try:
        q = int(rng.choice([1, 2, 3, 4, 5, 7, 10, 13]))
        sos, zi, padlen = _decimate_design(max(q, 2))
        L = int(rng.integers(padlen + 1, rng.choice([64, 400, 5000, 20000])  + padlen + 2))
        cols, nblocks = int(rng.integers(1, 12)), int(rng.integers(1, 8))
        t = np.arange(L * nblocks)[:, None]
        x = np.sin(0.01 * t * (1 + np.arange(cols))) * rng.uniform(0.1, 50) + rng.standard_normal((L * nblocks, cols)) * rng.uniform(0, 2) + rng.uniform(-5, 5)
        y = _filtfilt_device(x, nblocks, 0, sos[:, :3], sos[:, 3:], zi, padlen, q)
        ref = np.vstack([signal.sosfiltfilt(sos, x[b * L:(b + 1) * L], axis=0)[::q] for b in range(nblocks)])
        assert y.shape == ref.shape, (k, "shape", y.shape, ref.shape)
        assert np.abs(y - ref).max() <= 1e-13 * max(1e-300, np.abs(ref).max()), (k, "sosfiltfilt", L, cols, nblocks, q, np.abs(y - ref).max())
        if q >= 2:
            ref2 = np.vstack([signal.decimate(x[b * L:(b + 1) * L], q, zero_phase=True, axis=0) for b in range(nblocks)])
            assert np.abs(y - ref2).max() <= 1e-13 * np.abs(ref2).max(), (k, "decimate", L, cols, nblocks, q)
            # the joint-block helper of the scripts: W and tau decimated block by block
            tau = rng.standard_normal(L * nblocks)
            W_list, tau_list = decimate_joint_blocks(x, tau, nblocks, q=q, stages=1)
            for b in range(nblocks):
                assert np.abs(W_list[b] - ref2[b * len(W_list[b]):(b + 1) * len(W_list[b])]).max() <= 1e-13 * np.abs(ref2).max(), (k, "blocks")
                rt = signal.decimate(tau[b * L:(b + 1) * L], q, zero_phase=True)
                assert np.abs(tau_list[b] - rt).max() <= 1e-13 * max(1e-300, np.abs(rt).max()), (k, "tau block")
        nb = int(rng.choice([2, 3, 4, 5, 6]))
        Lf = int(rng.integers(3 * (nb + 1) + 10 * nb + 2, 6000))
        xf = np.cumsum(rng.standard_normal((Lf, int(rng.integers(1, 6)))), axis=0)
        ts = float(rng.choice([0.0002, 0.001, 0.01]))
        fc = float(rng.uniform(5, 0.4 / ts))
        param = {"ts": ts, "cut_off_frequency_butterworth": fc}
        got = low_pass_filter_data(xf, param, nb)
        b_, a_ = signal.butter(nb, ts * fc / 2, "low")
        reff = signal.filtfilt(b_, a_, xf, axis=0, padtype="odd", padlen=3 * (max(len(b_), len(a_)) - 1))[5 * nb:-5 * nb]
        assert got.shape == reff.shape and np.abs(got - reff).max() <= 1e-11 * max(1.0, np.abs(reff).max()), (k, "butter", nb, Lf, ts, fc)
    except Exception:  # noqa: BLE001
        bad += 1
        print("case", k, "FAILED")
        traceback.print_exc(limit=2)
print("%d cases, %d failures" % (cases, bad))
