# three workgroups per CU (48-row tiles, 168 VGPRs) against the shipped two for the human shape
export FIGH_LIB_PATH=$PWD/figaroh_plus_amd/libfigh_ab.so
for cfg in 4,3,4,2 4,3,3,3 4,3,3,2; do
  echo "== n=191 cfg=$cfg"; FIGH_WY_CFG=$cfg timeout 200 python tools/wide_tsqr_bench.py 4e6 191 2>&1 | grep "level0" | sed 's/|diag.*| level0/level0/'
done
echo "== n=175 cfg=4,3,3,3"; FIGH_WY_CFG=4,3,3,3 timeout 200 python tools/wide_tsqr_bench.py 4e6 175 2>&1 | grep "level0"
