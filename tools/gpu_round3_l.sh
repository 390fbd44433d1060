cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "merge or full_size_talos or full_size_tiago or pipeline_matches or human or sip or tls" 2>&1 | grep -E "passed|failed|error" | tail -3
python tools/wide_merge_bench.py 191 241 331 400 2>&1 | grep "count    1 \|count  512" | tee gpurun_out/r03/wide_merge_bench.txt
for c in cfg4; do
python bench.py --config $c --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$c', round(d['ms_per_step'],2), {k:(round(v.get('avg_ms'),3), v.get('launches', v.get('launches_per_step'))) for k,v in d['kernels'].items()}, d['config']['result_matches_reference'])
"
done
