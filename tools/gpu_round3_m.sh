cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "excitation or merge_tree or tsqr_shapes" 2>&1 | grep -E "passed|failed|error|Error|assert" | tail -8
timeout 600 python tools/objective_batch_bench.py > gpurun_out/r03/r03_objective_batch.json 2> gpurun_out/r03/objective_batch.err; tail -4 gpurun_out/r03/objective_batch.err
