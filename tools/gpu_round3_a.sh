cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "select_columns or merge_tree or merge_base or tsqr_selected or pipeline or tsqr_merge or base_parameters" 2>&1 | tail -25
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/b_r3b.json 2> gpurun_out/b_r3b.err; tail -c 1000 gpurun_out/b_r3b.err
python -c "
import json
d=json.load(open('gpurun_out/b_r3b.json'))
print(d['value'], d['ms_per_step'], {k:(v.get('avg_ms')) for k,v in d['kernels'].items()})
"
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_r3b -o r3b -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/b_r3b_prof.json 2>gpurun_out/b_r3b_prof.err
ls gpurun_out/prof_r3b | head
