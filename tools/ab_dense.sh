for rep in 1 2; do
  echo "== prev"; FIGH_LIB_PATH=$PWD/ab/libfigh_prev.so python tools/wide_tsqr_bench.py 4e6 191 241 331 2>&1 | grep level0 | sed "s/|diag.*| level0/level0/; s/| merges.*//"
  echo "== new";  python tools/wide_tsqr_bench.py 4e6 191 241 331 2>&1 | grep level0 | sed "s/|diag.*| level0/level0/; s/| merges.*//"
done
