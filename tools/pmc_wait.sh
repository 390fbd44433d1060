#!/bin/bash
# where the waves of a config's kernels spend their cycles: two rocprofv3 --pmc passes; usage: tools/pmc_wait.sh cfg4
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
c=${1:-cfg4}
O=gpurun_out/pmcw; mkdir -p $O
B="python3 bench.py --config $c --steps 2 --warmup 1 --no-cpu-baseline"
timeout -k 10 600 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM -d $O/${c}_a -o r -- $B > $O/${c}_a.log 2>&1
timeout -k 10 600 rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_INSTS_SALU SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS -d $O/${c}_b -o r -- $B > $O/${c}_b.log 2>&1
python3 tools/pmc_summary.py $O/$c.json $O/${c}_a/r_results.db $O/${c}_b/r_results.db > $O/$c.txt 2>&1
python3 - "$O/$c.json" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
rows = sorted(((v.get("duration_us_in_pmc_pass", 0), k, v) for k, v in d.items()), reverse=True)[:6]
for dur, k, v in rows:
    wc = v.get("SQ_WAVE_CYCLES", 0) or 1
    print("%-52s %9.1f us" % (k[:52], dur))
    print("    of wave cycles: wait any %.2f, wait inst %.2f, active any %.2f, VALU %.2f, LDS %.2f, VMEM %.2f, SCA %.2f, MISC %.2f, wait LDS %.2f; MFMA busy / SQ busy %.2f" % (tuple(
        v.get(n, 0) / wc for n in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_VMEM", "SQ_ACTIVE_INST_SCA", "SQ_ACTIVE_INST_MISC", "SQ_WAIT_INST_LDS")) + (v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (v.get("SQ_BUSY_CYCLES", 0) or 1) / 4,)))
PY
find $O -name "*.db" -size +20M -delete
