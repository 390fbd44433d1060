#!/bin/bash
# Same-box A/B of library builds on the step of a config: tools/ab_step.sh <config> <repetitions> <lib>...  (GPU box scratch copy;
# every <lib> is copied over figaroh_plus_amd/libfigh.so in turn -- alternating, so that box drift shows -- and put back at the end)
cfg=$1; reps=$2; shift 2
keep=$(mktemp /tmp/libfigh_shipped.XXXXXX.so); cp figaroh_plus_amd/libfigh.so $keep
trap 'cp $keep figaroh_plus_amd/libfigh.so; rm -f $keep' EXIT
for r in $(seq $reps); do for lib in "$@"; do
  [ "$lib" = shipped ] && cp $keep figaroh_plus_amd/libfigh.so || cp $lib figaroh_plus_amd/libfigh.so
  python bench.py --config $cfg --steps 6 --warmup 2 --no-cpu-baseline --strong-config= 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
k=d['kernels']
print('$lib', '$cfg', 'step %.3f ms' % d['ms_per_step'], ' '.join('%s %.3f x %.1f' % (n, v['avg_ms'], v.get('launches', 0) / 6.0) for n, v in k.items() if 'launches' in v), 'ok' if d['config']['result_matches_reference'] else 'MISMATCH')"
done; done
