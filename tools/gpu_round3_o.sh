cd $GRAFT_REPO_ROOT
bash tools/wyoff_ab.sh
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "merge or human or streamed or chunked" 2>&1 | grep -E "passed|failed|error" | tail -3
python bench.py --config cfg5 --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('cfg5', round(d['ms_per_step'],2), {k:(round(v.get('avg_ms'),3), v.get('launches', v.get('launches_per_step'))) for k,v in d['kernels'].items()}, d['config']['result_matches_reference'])
"
