# same-box A/B of two library builds on a BASELINE config: ab/libfigh_prev.so against the tree's libfigh.so
cfg=${1:-cfg4}
for rep in 1 2; do
  echo "== prev"; FIGH_LIB_PATH=$PWD/ab/libfigh_prev.so python tools/step_profile.py $cfg 4 2>&1 | grep "ms per step\|kernel averages"
  echo "== new";  python tools/step_profile.py $cfg 4 2>&1 | grep "ms per step\|kernel averages"
done
