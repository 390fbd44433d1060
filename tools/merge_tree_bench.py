#!/usr/bin/env python3
"""Time the one-launch merge tree (figh_tsqr_tree.hip) per number of levels: stacks of 1 / 15 / 133 / 2039 triangles of
nc columns through figh_tsqr_merge and figh_tsqr_merge_base (HIP events of the library, 'tsqr_tree' scope)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from figaroh_plus_amd import _lib as lib  # noqa: E402

nc = int(sys.argv[1]) if len(sys.argv) > 1 else 50
rng = np.random.default_rng(0)
lib.load()
for count in (1, 2, 9, 15, 16, 133, 230, 2039):
    stack = np.triu(rng.standard_normal((count, nc, nc)))
    d_stack = lib.DeviceArray.from_host(stack.reshape(-1))
    d_R = lib.DeviceArray(((nc + 1) * nc,))
    for name, fn in (("merge", lambda: lib.tsqr_merge(d_stack, count, nc, d_R)),
                     ("merge_base", lambda: lib.tsqr_merge_base(d_stack, count, nc, nc - 1, 1e-8, d_R))):
        if name == "merge" and count == 1:
            continue
        for _ in range(3):
            fn()
        lib.profile_enable(True, level=2)
        lib.profile_reset()
        for _ in range(20):
            fn()
        cnt, ms = lib.profile_get("tsqr_tree")
        lib.profile_enable(False)
        print("nc %d count %5d %-10s launches %d avg %.1f us" % (nc, count, name, cnt, 1e3 * ms / max(cnt, 1)))
