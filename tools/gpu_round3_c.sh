cd $GRAFT_REPO_ROOT
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/b_r3c.json 2> gpurun_out/b_r3c.err; tail -c 600 gpurun_out/b_r3c.err
python -c "
import json
d=json.load(open('gpurun_out/b_r3c.json'))
print(d['value'], d['ms_per_step'], {k:(v.get('avg_ms')) for k,v in d['kernels'].items()})
"
timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -15
