#!/bin/bash
# one rocprofv3 --pmc pass of the default bench command; usage: tools/pmc_quick.sh <tag> COUNTER [COUNTER ...]
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
tag=$1; shift
O=gpurun_out/pmcq; mkdir -p $O
timeout -k 10 300 rocprofv3 --kernel-trace --pmc "$@" -d $O/$tag -o r -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline > $O/$tag.log 2>&1
python3 tools/pmc_summary.py $O/$tag.json $O/$tag/r_results.db 2>&1 | head -4
find $O -name "*.db" -size +20M -delete
