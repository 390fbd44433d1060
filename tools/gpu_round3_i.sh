cd $GRAFT_REPO_ROOT
for r in 1 2 3; do
python bench.py --steps 20 --warmup 5 --no-cpu-baseline | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print(round(d['ms_per_step'],4), d['config']['w_placement'], {k:round(v.get('avg_ms'),4) for k,v in d['kernels'].items()})
"
done
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "pipeline or full_size_talos or bench_two" 2>&1 | tail -3
