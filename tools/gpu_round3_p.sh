cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "wrench_force or human or streamed or chunked or sip or gram" 2>&1 | grep -E "passed|failed|error|Error|assert|^E " | tail -6
for c in cfg5; do
python bench.py --config $c --steps 3 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$c', round(d['ms_per_step'],3), {k:(round(v.get('avg_ms'),3), v.get('launches', v.get('launches_per_step'))) for k,v in d['kernels'].items()}, d['config']['result_matches_reference'])
"
done
