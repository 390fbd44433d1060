# 80-row tiles (NRC = 5) against 64-row tiles for up to 12 chunks (the human shape), ablation build
export FIGH_LIB_PATH=$PWD/figaroh_plus_amd/libfigh_ab.so
for rep in 1 2; do for cfg in 4,3,4,2 4,3,5,2; do
  echo "== cfg=$cfg"; FIGH_WY_CFG=$cfg timeout 200 python tools/wide_tsqr_bench.py 4e6 191 150 2>&1 | grep "level0" | sed 's/| merges.*//'
done; done
