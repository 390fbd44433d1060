# which SIMD hosts the owner of a panel, relative to the other workgroup of the CU (ablation build, FIGH_WY_OFF)
export FIGH_LIB_PATH=$PWD/figaroh_plus_amd/libfigh_ab.so
for n in 191 331; do
  for off in 0 1 2 3 4 0; do
    echo "== n=$n off=$off $(FIGH_WY_OFF=$off timeout 200 python tools/wide_tsqr_bench.py 4e6 $n 2>&1 | grep level0 | sed 's/.*| level0/level0/; s/| merges.*//')"
  done
done
