"""K1' alone (figh_regressor_build on a tree model): kernel time by HIP events and algorithmic GB/s.
usage: python tools/tree_kernel_bench.py [scale]   (FIGH_LIB_PATH / FIGH_TREE_TAPE select ablation variants)"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
from figaroh_plus_amd import _lib
from figaroh_plus_amd.device import GpuMatrix
from figaroh_plus_amd.tools.regressor import regressor_flags
from figaroh_plus_amd.tools.robot import Robot
from figaroh_plus_amd.tools.randomdata import sample_inputs  # noqa

scale = float(sys.argv[1]) if len(sys.argv) > 1 else 0.25
for cfg, mn, N in (("cfg3_tiago", "tiago", 1_000_000), ("cfg4_talos", "talos", 4_000_000), ("cfg5_human", "human", 2_000_000)):
    N = int(N * scale)
    meta = json.load(open(os.path.join(ROOT, "tests", "golden", cfg + ".json")))
    robot = Robot.from_flat(mn)
    q, v, a = sample_inputs(robot.model, N, np.random.default_rng(5), 1.5, 2, 5)
    d_q, d_v, d_a = (_lib.DeviceArray.from_host(np.ascontiguousarray(x).reshape(-1)) for x in (q, v, a))
    mode, flags, ft = regressor_flags(meta["param"], meta["coupling"])
    h = robot.device_model()
    rps, ncols = h.shape(mode, flags)
    padded = not os.environ.get("FIGH_BENCH_DENSE")
    W = GpuMatrix.empty(rps * N, 16 * (robot.model.njoints - 1) if padded else ncols)
    build = _lib.regressor_build_padded if padded else _lib.regressor_build
    d_c = _lib.DeviceArray((ncols,), np.float64)
    for _ in range(2):
        build(h, mode, flags, ft, N, d_q, d_v, d_a, W.buf, W.ld, d_c)
    _lib.synchronize(); _lib.profile_enable(True); _lib.profile_reset()
    K = 5
    for _ in range(K):
        build(h, mode, flags, ft, N, d_q, d_v, d_a, W.buf, W.ld, d_c)
    cnt, ms = _lib.profile_get("regressor_tree"); _lib.profile_enable(False)
    ms /= max(cnt, 1)
    byt = N * (8 * (robot.model.nq + 2 * robot.model.nv) + 8 * rps * ncols)
    print("%-11s N=%d  K1' %.2f ms  %.0f GB/s algorithmic (%.1f%% of 8 TB/s)  layout=%s tape=%s" % (
        cfg, N, ms, byt / ms / 1e6, 100 * byt / ms / 1e6 / 8000, "link-padded" if padded else "dense (reference)",
        os.environ.get("FIGH_TREE_TAPE")), flush=True)
    del W, d_q, d_v, d_a
