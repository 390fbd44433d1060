# same-box A/B: ab/libfigh_prev.so (tools/build_prev.sh) against the in-tree library, three alternations
cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do
  FIGH_LIB_PATH=$PWD/ab/libfigh_prev.so python tools/ab_step.py
  python tools/ab_step.py
done
