#!/bin/bash
# LDS bank conflicts per kernel of a bench config: one rocprofv3 --pmc pass; usage: tools/pmc_lds.sh cfg3|cfg4|cfg5
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
c=${1:-cfg4}
O=gpurun_out/pmcl; mkdir -p $O
timeout -k 10 600 rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES -d $O/$c -o r -- python3 bench.py --config $c --steps 2 --warmup 1 --no-cpu-baseline > $O/$c.log 2>&1
python3 tools/pmc_summary.py $O/$c.json $O/$c/r_results.db > $O/$c.txt 2>&1
python3 - "$O/$c.json" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
rows = []
for k, v in d.items():
    if "SQ_INSTS_LDS" in v and v.get("SQ_INSTS_LDS", 0) > 0:
        rows.append((v.get("duration_us_in_pmc_pass", 0), k, v["SQ_INSTS_LDS"], v.get("SQ_LDS_BANK_CONFLICT", 0), v.get("SQ_LDS_IDX_ACTIVE", 0), v.get("SQ_BUSY_CYCLES", 0)))
for dur, k, n, bc, act, busy in sorted(rows, reverse=True)[:14]:
    print("%-58s %9.1f us  LDS insts %.3g  conflict cycles %.3g  idx active %.3g  busy %.3g  conflict/active %.2f" % (k[:58], dur, n, bc, act, busy, bc / act if act else 0))
PY
find $O -name "*.db" -size +20M -delete
