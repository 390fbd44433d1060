cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "repack or blocked or streamed or chunked or pipeline_matches or full_size_talos or handwritten" 2>&1 | tail -4
for c in cfg3 cfg4 cfg5; do
python bench.py --config $c --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$c', round(d['ms_per_step'],2), {k:round(v.get('avg_ms'),3) for k,v in d['kernels'].items()}, d['transfers'].get('repack_inputs_ms'), d['config']['result_matches_reference'])
"
done
