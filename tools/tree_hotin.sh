# K1' with the sample inputs served from cache (ablation build, FIGH_TREE_HOTIN) against the real kernel: how much of the
# regressor pass is the strided, line-granular re-reading of q, v, a
export FIGH_LIB_PATH=$PWD/figaroh_plus_amd/libfigh_ab.so
for cfg in cfg4 cfg5 cfg3; do
  echo "== $cfg real";   python tools/step_profile.py $cfg 2 2>&1 | grep "kernel averages"
  echo "== $cfg hot inputs"; FIGH_TREE_HOTIN=1 python tools/step_profile.py $cfg 2 2>&1 | grep "kernel averages"
done
