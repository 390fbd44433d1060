export FIGH_LIB_PATH=$PWD/figaroh_plus_amd/libfigh_ab.so
run() { echo "== cfg=$1 n=$2"; FIGH_WY_CFG=$1 FIGH_WY_PROF=1 timeout 200 python tools/wide_tsqr_bench.py 2e6 $2 2>&1 | grep -v "^device" | sed 's/\[wy prof\]/  prof/' | awk '/prof/ && !seen[$0]++ && ++n<=1 {print} !/prof/ {print}'; }
run 4,3,4,2 191
run 4,4,4,2 241; run 4,4,3,2 241
run 8,3,4,2 331; run 4,6,3,2 331; run 4,6,4,1 331
run 8,4,4,2 400
