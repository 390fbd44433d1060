#!/bin/bash
# cfg5 (human, 1e7 samples, streamed): step time against the chunk size of the streamed pass
for c in ${CHUNKS:-500000 1000000 2500000 5000000}; do
  python bench.py --config cfg5 --chunk-samples $c --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 > /tmp/chunk_$c.json
  python - "$c" <<'PY'
import json, sys
c = sys.argv[1]
try:
    d = json.load(open("/tmp/chunk_%s.json" % c))
    print(c, round(d["ms_per_step"], 2), d["config"]["result_matches_reference"], {k: round(v.get("avg_ms", 0), 3) for k, v in d["kernels"].items()})
except Exception as e:
    print(c, "failed:", e)
PY
done
