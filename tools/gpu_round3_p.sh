cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "tsqr_shapes or ragged_rows or randomised or merge_tree or tree_row_blocks or structured_tsqr or excitation" 2>&1 | grep -E "passed|failed|error|Error|assert|^E " | tail -6
python tools/wide_tsqr_bench.py 4e6 86 96 2>&1 | grep level0 | sed 's/.*| level0/level0/; s/| wall.*//'
for c in cfg3; do
python bench.py --config $c --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$c', round(d['ms_per_step'],3), {k:(round(v.get('avg_ms'),3), v.get('launches', v.get('launches_per_step'))) for k,v in d['kernels'].items()}, d['config']['result_matches_reference'])
"
done
