cd $GRAFT_REPO_ROOT
python tools/merge_tree_bench.py 50 | grep -E "count +(1|15|133|2039) "
python tools/ab_step.py; python tools/ab_step.py
timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -6
