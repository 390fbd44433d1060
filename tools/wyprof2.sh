export FIGH_LIB_PATH=$PWD/figaroh_plus_amd/libfigh_ab.so
for cfg in 4,6,3,2 4,5,4,2,1; do
FIGH_WY_CFG=$cfg FIGH_WY_PROF=1 timeout 200 python tools/wide_tsqr_bench.py 2e6 331 2>&1 | grep -v "^device" | awk '/prof/ && !seen[$0]++ && ++n<=1 {print} !/prof/ {print}'
done
