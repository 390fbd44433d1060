#!/usr/bin/env python3
"""Time the merge levels of the blocked (nc > 80) TSQR: stacks of 2 / 8 / 64 / 512 triangles through figh_tsqr_merge and
figh_tsqr_merge_base (wall clock around synchronised calls, median of 10)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from figaroh_plus_amd import _lib as lib  # noqa: E402

rng = np.random.default_rng(0)
lib.load()
for nc in [int(a) for a in sys.argv[1:]] or [191, 241, 331]:
    for count in (1, 2, 8, 64, 512):
        stack = np.triu(rng.standard_normal((count, nc, nc)))
        d_stack = lib.DeviceArray.from_host(stack.reshape(-1))
        d_R = lib.DeviceArray(((nc + 1) * nc,))
        for name, fn in (("merge", lambda: lib.tsqr_merge(d_stack, count, nc, d_R)),
                         ("merge_base", lambda: lib.tsqr_merge_base(d_stack, count, nc, nc - 1, 1e-8, d_R))):
            if name == "merge" and count == 1:
                continue
            ts = []
            for _ in range(12):
                lib.synchronize()
                t0 = time.perf_counter()
                fn()
                lib.synchronize()
                ts.append(time.perf_counter() - t0)
            print("nc %d count %4d %-10s median %.3f ms" % (nc, count, name, 1e3 * float(np.median(ts[2:]))), flush=True)
