export FIGH_LIB_PATH=$PWD/figaroh_plus_amd/libfigh_ab.so
for rep in 1 2; do for cfg in 8,4,4,2 4,6,3,2,1; do
  echo "== n=400 cfg=$cfg"; FIGH_WY_CFG=$cfg timeout 200 python tools/wide_tsqr_bench.py 4e6 400 2>&1 | grep "level0" | sed 's/| merges.*//'
done; done
