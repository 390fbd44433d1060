#!/bin/bash
# A/B of libfigh builds on a GPU box (a scratch copy of the repo): tools/var_run.sh <config> <lib>...  -- every <lib> is copied
# over figaroh_plus_amd/libfigh.so in turn, bench.py is run on it, the shipped library is put back at the end.
cfg=$1; shift
keep=$(mktemp /tmp/libfigh_shipped.XXXXXX.so); cp figaroh_plus_amd/libfigh.so $keep
trap 'cp $keep figaroh_plus_amd/libfigh.so; rm -f $keep' EXIT  # (also on a failing or interrupted run)
for lib in "$@"; do
  cp $lib figaroh_plus_amd/libfigh.so
  python bench.py --config $cfg --steps 6 --warmup 2 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
k=d['kernels']['regressor_tree']
print('$lib', '$cfg', 'step %.3f ms' % d['ms_per_step'], 'K1 %.3f ms' % k['avg_ms'], 'frac %.3f' % k['frac'])"
done
