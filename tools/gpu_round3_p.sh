cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "structural_zeros or tree_row_blocks or regressor or pipeline or full_size_tiago" 2>&1 | grep -E "passed|failed|error|Error|assert|^E " | tail -8
for z in dense auto; do
python bench.py --config cfg3 --steps 5 --warmup 2 --no-cpu-baseline --w-layout $z 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('cfg3', '$z', round(d['ms_per_step'],3), {k:(round(v.get('avg_ms'),3), v.get('launches', v.get('launches_per_step'))) for k,v in d['kernels'].items()}, d['config']['result_matches_reference'], d['config']['w_layout'])
"
done
