export FIGH_LIB_PATH=$PWD/figaroh_plus_amd/libfigh_ab.so
for rep in 1 2; do
for cfg in 4,3,5,2 4,3,6,2; do echo "== n=191 cfg=$cfg"; FIGH_WY_CFG=$cfg timeout 200 python tools/wide_tsqr_bench.py 4e6 191 2>&1 | grep "level0" | sed 's/| merges.*//'; done
for cfg in 4,4,4,2 4,4,5,2; do echo "== n=241 cfg=$cfg"; FIGH_WY_CFG=$cfg timeout 200 python tools/wide_tsqr_bench.py 4e6 241 2>&1 | grep "level0" | sed 's/| merges.*//'; done
done
