"""In-kernel step profile of the streamed merge tree (ablation build libfigh_ab.so, `make -C figaroh_plus_amd/csrc ab`):
s_memtime ticks per column step of wave 0 / workgroup 0, split at five points.
usage: FIGH_LIB_PATH=figaroh_plus_amd/libfigh_ab.so python tools/stream_prof.py [count] [nc]"""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from figaroh_plus_amd import _lib as lib
count = int(sys.argv[1]) if len(sys.argv) > 1 else 2039
nc = int(sys.argv[2]) if len(sys.argv) > 2 else 50
L = lib.load()
raw = C.CDLL(lib.LIB_PATH)
rng = np.random.default_rng(0)
d_stack = lib.DeviceArray.from_host(np.triu(rng.standard_normal((count, nc, nc))).reshape(-1))
d_R = lib.DeviceArray(((nc + 1) * nc,))
buf = (C.c_longlong * 8)()
for _ in range(3):
    lib.tsqr_merge_base(d_stack, count, nc, nc - 1, 1e-8, d_R)
lib.synchronize(); raw.figh_ab_stream_prof(buf, 1)
reps = 20
for _ in range(reps):
    lib.tsqr_merge_base(d_stack, count, nc, nc - 1, 1e-8, d_R)
lib.synchronize(); raw.figh_ab_stream_prof(buf, 1)
names = ["arrival (request next + validate)", "dots + row-group reduce + LDS write", "barrier", "sums + rsq chain", "update + row store"]
tot = sum(buf[i] for i in range(5))
print("count %d nc %d: %.0f ticks per step (wave 0 of workgroup 0, %d steps x %d launches)" % (count, nc, tot / (reps * nc), nc, reps))
for i, n in enumerate(names):
    print("  %-40s %8.0f ticks per step  %5.1f %%" % (n, buf[i] / (reps * nc), 100.0 * buf[i] / max(tot, 1)))
