#!/bin/bash
# step time and pivot gap of every bench config with and without the null-pivot rule
mkdir -p gpurun_out/r04
for c in ${CONFIGS:-cfg2 cfg3 cfg4 cfg5}; do
  for f in "" "--no-null-pivots"; do
    echo "== $c $f"
    python bench.py --config $c $f --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['config']['pivots'], d['config']['result_matches_reference'], {k:round(v.get('avg_ms',0),3) for k,v in d['kernels'].items()})"
  done
done
