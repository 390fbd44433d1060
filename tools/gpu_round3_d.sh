cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python tools/step_trace.py cfg2 3 2>&1 | tail -30
sed -i 's/LAST=(rows\[-1\]\[1\]-t0)\/1e6-130/LAST=(rows[-1][1]-t0)\/1e6-5/' tools/trace_gaps.sh
bash tools/trace_gaps.sh cfg2 2>&1 | tail -30
