# chained level-0 launches of the streamed human pass against one stack of triangles per chunk, same box (ablation build)
export FIGH_LIB_PATH=$PWD/figaroh_plus_amd/libfigh_ab.so
cat > /tmp/chain_ab.py <<'PY'
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.getcwd())
from figaroh_plus_amd import _lib
from figaroh_plus_amd.pipeline import IdentificationPipeline
from figaroh_plus_amd.tools.randomdata import sample_inputs
from figaroh_plus_amd.tools.robot import Robot
meta = json.load(open("tests/golden/cfg5_human.json"))
robot = Robot.from_flat("human")
N = 4_000_000
q, v, a = sample_inputs(robot.model, N, np.random.default_rng(5), 1.5, 2, 5)
pipe = IdentificationPipeline(robot, meta["param"], params_std=dict(zip(meta["names_std"], meta["phi_ref_raw"])), chunk_samples=500000)
pipe.set_samples(q, v, a)
pipe.set_tau_from_parameters(np.array([float(x) for x in meta["phi_ref_raw"]]))
pipe.run()
_lib.profile_enable(True, level=2); _lib.profile_reset()
ts = []
for _ in range(3):
    t0 = time.perf_counter(); out = pipe.run(); ts.append(time.perf_counter() - t0)
print(os.environ.get("FIGH_NO_CHAIN", "chained"), "step ms", [round(1e3 * t, 1) for t in ts], {k: (_lib.profile_get(k)[0], round(_lib.profile_get(k)[1] / max(_lib.profile_get(k)[0], 1), 3)) for k in ("tsqr", "regressor_tree", "tsqr_reduce")}, len(out["idx_base"]))
PY
for r in 1; do
  timeout 300 python /tmp/chain_ab.py
  FIGH_NO_CHAIN=1 timeout 300 python /tmp/chain_ab.py
  FIGH_CHAIN_MASK=1 timeout 300 python /tmp/chain_ab.py 2>&1 | tail -1
  FIGH_CHAIN_MASK=2 timeout 300 python /tmp/chain_ab.py 2>&1 | tail -1
  FIGH_CHAIN_MASK=0 timeout 300 python /tmp/chain_ab.py 2>&1 | tail -1
done
