cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | grep -E "passed|failed|error" | tail -3 | tee gpurun_out/r03/r03_gputest_summary.txt
timeout 1200 python tools/shard_sweep.py cfg2 cfg3 cfg4 cfg5 > gpurun_out/r03/r03_shard_sweep.json 2> gpurun_out/r03/shard.err; tail -4 gpurun_out/r03/shard.err
