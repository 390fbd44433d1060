# the narrow TSQR of this tree against the round-1 library (ab/libfigh_r01.so, built from the round-1 commit) on one box
for rep in 1 2 3; do
  FIGH_OLD_ABI=1 FIGH_LIB_PATH=$PWD/ab/libfigh_r01.so python tools/narrow_ab.py 2>&1 | grep " tsqr \| regressor_chain"
  python tools/narrow_ab.py 2>&1 | grep " tsqr \| regressor_chain"
done
