#!/usr/bin/env python3
"""Extended run of tests/test_gpu_parity.py::test_tree_walks_fuzz (random trees through both walks of the tape kernel against the C
oracle) over many more seeds than the suite's sixteen.  usage: python tools/fuzz_trees.py [first_seed] [count]"""
import os
import sys
import traceback

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p)
import oracle_c  # noqa: E402  (checker only)
import test_gpu_parity as T  # noqa: E402
from figaroh_plus_amd import _lib  # noqa: E402

_lib.load()
first = int(sys.argv[1]) if len(sys.argv) > 1 else 100
count = int(sys.argv[2]) if len(sys.argv) > 2 else 200
fn = getattr(T.test_tree_walks_fuzz, "__wrapped__", T.test_tree_walks_fuzz)
bad = 0
for seed in range(first, first + count):
    try:
        fn(_lib, oracle_c, seed)
    except Exception:  # noqa: BLE001
        bad += 1
        print("seed", seed, "FAILED")
        traceback.print_exc(limit=3)
print("%d random trees (seeds %d .. %d), %d failures" % (count, first, first + count - 1, bad))
