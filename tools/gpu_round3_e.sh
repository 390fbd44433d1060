cd $GRAFT_REPO_ROOT
for r in 1 2; do FIGH_LIB_PATH=$PWD/ab/libfigh_prev.so python tools/k1_alloc_probe.py | sed "s/.*K1 \([0-9.]*\) ms.*memset \([0-9.]*\).*/prev \1 \2/"; python tools/k1_alloc_probe.py | sed "s/.*K1 \([0-9.]*\) ms.*memset \([0-9.]*\).*/new \1 \2/"; done | paste - - - - - -
for d in d1 new d3 d4; do
  if [ $d = new ]; then unset FIGH_LIB_PATH; else export FIGH_LIB_PATH=$PWD/ab/libfigh_$d.so; fi
  echo "== depth $d"; python tools/merge_tree_bench.py 50 | grep -E "count +(1|15|133|2039) "
done
unset FIGH_LIB_PATH
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "merge_tree or merge_base or tsqr_selected or pipeline" 2>&1 | tail -3
