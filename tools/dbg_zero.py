import sys, os, json
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np
from conftest import Golden
from figaroh_plus_amd.tools.regressor import build_regressor_basic
for cfg in ["cfg3_tiago", "cfg4_talos", "cfg5_human"]:
    g = Golden(cfg)
    robot = g.robot()
    W = build_regressor_basic(robot, g["q_small"], g["v_small"], g["a_small"], g.param)
    ref = g["W_small"]
    bad = np.argwhere((W == 0) != (ref == 0))
    print(cfg, "mismatches", len(bad), "maxerr", np.abs(W - ref).max() / np.abs(ref).max())
    N = len(g["q_small"])
    seen = set()
    for r, c in bad[:2000]:
        key = (r // N, c // 14, c % 14)
        if key in seen: continue
        seen.add(key)
        if len(seen) < 25: print("  rowblock", r // N, "link", c // 14 + 1, "slot", c % 14, "W", W[r, c], "ref", ref[r, c])
    print("  distinct (rowblock, link, slot):", len(seen))
