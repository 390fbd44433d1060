export FIGH_LIB_PATH=$PWD/figaroh_plus_amd/libfigh_ab.so
FIGH_WY_CFG=4,3,4,1 FIGH_WY_PROF=1 timeout 200 python tools/wide_tsqr_bench.py 2e6 191 2>&1 | grep -v "^device" | awk '/prof/ && !seen[$0]++ && ++n<=1 {print} !/prof/ {print}'
FIGH_WY_CFG=4,6,4,1 FIGH_WY_PROF=1 timeout 200 python tools/wide_tsqr_bench.py 2e6 331 2>&1 | grep -v "^device" | awk '/prof/ && !seen[$0]++ && ++n<=1 {print} !/prof/ {print}'
