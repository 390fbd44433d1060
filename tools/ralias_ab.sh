# R blocks from L2 (aliased, garbage results) against the real kernel: how much of the wide TSQR is R traffic
export FIGH_LIB_PATH=$PWD/figaroh_plus_amd/libfigh_ab.so
for n in 191 241 331; do
  echo "== n=$n real";  timeout 200 python tools/wide_tsqr_bench.py 4e6 $n 2>&1 | grep "level0" | sed 's/|diag.*| level0/level0/'
  echo "== n=$n aliased R"; FIGH_WY_RALIAS=1 timeout 200 python tools/wide_tsqr_bench.py 4e6 $n 2>&1 | grep "level0" | sed 's/|diag.*| level0/level0/'
done
