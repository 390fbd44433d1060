cd $GRAFT_REPO_ROOT
for r in 1 2 3; do FIGH_LIB_PATH=$PWD/ab/libfigh_prev.so python tools/k1_alloc_probe.py | sed "s/.*K1 \([0-9.]*\) ms.*memset \([0-9.]*\).*/prev \1 \2/"; python tools/k1_alloc_probe.py | sed "s/.*K1 \([0-9.]*\) ms.*memset \([0-9.]*\).*/new \1 \2/"; done | paste - - - - - -
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "regressor or elimination or tx40 or leading_dimension or full_size_structural" 2>&1 | tail -3
