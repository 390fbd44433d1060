export FIGH_LIB_PATH=$PWD/figaroh_plus_amd/libfigh_ab.so
for t in zrun2 zrun3 zrun4 zrun6 zrun8; do
  export FIGH_TREE_TAPE=$t
  timeout 300 python tools/tree_kernel_bench.py 0.25 2>&1 | tail -2
done
