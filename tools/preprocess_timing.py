"""two decimate-by-10 stages over every column of a TX40-sized regressor (6 x 44958 rows x 87 columns)
and tau, device kernel against scipy.signal.decimate on the host cores; plus the Butterworth filtfilt of q."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle")]
import numpy as np
import oracle_np
from figaroh_plus_amd import _lib
from figaroh_plus_amd.identification import identification_tools as idt

nj, ncols = 44958, 87
rng = np.random.default_rng(0)
t = np.arange(6 * nj)[:, None]
W = np.sin(1e-3 * t * (1 + np.arange(ncols))) + 0.05 * rng.standard_normal((6 * nj, ncols))
tau = W @ rng.standard_normal(ncols)
for name, fn in (("device", idt.decimate_joint_blocks), ("scipy ", oracle_np.decimate_joint_blocks)):
    fn(W, tau, 6)
    t0 = time.perf_counter(); Wl, tl = fn(W, tau, 6); dt = time.perf_counter() - t0
    print("%s decimate 2 x q=10 of %d x %d (+tau): %.1f ms  -> blocks of %d rows" % (name, W.shape[0], ncols, dt * 1e3, Wl[0].shape[0]))
    if name == "device":
        ref = (Wl, tl)
err = max(np.abs(a - b).max() for a, b in zip(ref[0], Wl))
print("max |device - scipy| = %.2e" % err)
_lib.profile_enable(True, level=2); _lib.profile_reset()
idt.decimate_joint_blocks(W, tau, 6)
n, ms = _lib.profile_get("filtfilt_cols")
print("kernel time inside the device path: %d launches, %.2f ms total (the rest is PCIe + numpy slicing)" % (n, ms))
q = np.cumsum(rng.standard_normal((44998, 6)), axis=0) * 1e-3
param = {"ts": 0.0002, "cut_off_frequency_butterworth": 100.0}
for name, fn in (("device", idt.low_pass_filter_data), ("scipy ", oracle_np.low_pass_filter_data)):
    fn(q, param, 4)
    t0 = time.perf_counter(); out = fn(q, param, 4); dt = time.perf_counter() - t0
    print("%s filtfilt(butter 4) of %d x 6: %.2f ms" % (name, q.shape[0], dt * 1e3))
