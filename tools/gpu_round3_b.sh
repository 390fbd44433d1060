cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "merge_tree or merge_base or tsqr_selected or pipeline or tsqr_merge or base_parameters" 2>&1 | tail -25
timeout 120 python tools/merge_tree_bench.py 50
