# same-box A/B of two library builds on the UR10 pass: merge-level time and step time
for rep in 1 2 3; do
  for lib in prev new; do
    if [ $lib = prev ]; then export FIGH_LIB_PATH=$PWD/ab/libfigh_prev.so; else unset FIGH_LIB_PATH; fi
    python - <<'PY'
import os, sys, time, json
sys.path.insert(0, os.getcwd())
import numpy as np
from figaroh_plus_amd import _lib
from figaroh_plus_amd.pipeline import IdentificationPipeline
from figaroh_plus_amd.tools.robot import Robot
meta = json.load(open("tests/golden/cfg2_ur10.json"))
robot = Robot.from_flat("ur10"); N = 1_000_000
rng = np.random.default_rng(1); q, v, a = (rng.uniform(-6, 6, (N, 6)) for _ in range(3))
pipe = IdentificationPipeline(robot, meta["param"], params_std=dict(zip(meta["names_std"], meta["phi_ref_raw"])))
pipe.set_samples(q, v, a); pipe.set_tau_from_parameters(np.array([float(x) for x in meta["phi_ref_raw"]]), noise_std=0.05, seed=0)
for _ in range(3): out = pipe.run()
_lib.synchronize(); t0 = time.perf_counter()
for _ in range(30): out = pipe.run()
_lib.synchronize(); step = (time.perf_counter() - t0) / 30
_lib.profile_enable(True, level=2); _lib.profile_reset()
for _ in range(5): pipe.run()
c, ms = _lib.profile_get("tsqr_reduce")
print(os.path.basename(_lib.LIB_PATH), "step %.3f ms  merges %.3f ms per step (%d launches)  base %d" % (1e3 * step, ms / 5, c // 5, len(out["idx_base"])))
PY
  done
done
