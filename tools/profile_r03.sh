#!/bin/bash
# Round-3 profiles on the GPU box: kernel stats and PMC passes of the bench command (cfg2) and of the TALOS / TIAGo / human
# configs.  Results under gpurun_out/r03/ (copied to profiles/ by hand).  rocprofv3 is given the program itself
# (python3 bench.py ...), counters in their own passes.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03; mkdir -p $O
SEL=${1:-"cfg2 cfg3 cfg4 cfg5"}
want() { case " $SEL " in *" $1 "*) return 0;; *) return 1;; esac; }
rocprofv3 --list-avail 2>/dev/null | grep -o "SQ_INSTS_VALU_MFMA[A-Z0-9_]*\|SQ_VALU_MFMA_BUSY_CYCLES\|SQ_INSTS_MFMA\|SQ_BUSY_CYCLES\|SQ_INSTS_VALU\b" | sort -u > $O/avail_counters.txt
prof() { # name, extra rocprof args..., -- command
  local name=$1; shift
  timeout -k 10 300 rocprofv3 --kernel-trace "$@" > $O/$name.log 2>&1  # (bounded: a counter pass that dies can hang in finalisation)
}
B2="python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline"
B4="python3 bench.py --config cfg4 --steps 3 --warmup 1"
B3="python3 bench.py --config cfg3 --steps 3 --warmup 1"
B5="python3 bench.py --config cfg5 --steps 2 --warmup 1"
want cfg2 && { prof cfg2_stats --stats -d $O/cfg2_stats -o r -- $B2;  grep '^{' $O/cfg2_stats.log | tail -1 > $O/r03_cfg2_bench_under_rocprof.json; }
want cfg2 && prof cfg2_fetch --pmc FETCH_SIZE -d $O/cfg2_fetch -o r -- $B2
want cfg2 && prof cfg2_write --pmc WRITE_SIZE -d $O/cfg2_write -o r -- $B2
want cfg2 && prof cfg2_sq --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_BUSY_CYCLES SQ_WAVES -d $O/cfg2_sq -o r -- $B2
want cfg4 && { prof cfg4_stats --stats -d $O/cfg4_stats -o r -- $B4;  grep '^{' $O/cfg4_stats.log | tail -1 > $O/r03_cfg4_bench_under_rocprof.json; }
want cfg4 && prof cfg4_fetch --pmc FETCH_SIZE -d $O/cfg4_fetch -o r -- $B4
want cfg4 && prof cfg4_write --pmc WRITE_SIZE -d $O/cfg4_write -o r -- $B4
want cfg4 && prof cfg4_mfma --pmc SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_INSTS_VALU SQ_BUSY_CYCLES SQ_WAVES -d $O/cfg4_mfma -o r -- $B4
want cfg4 && prof cfg4_mfma2 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_BUSY_CU_CYCLES -d $O/cfg4_mfma2 -o r -- $B4
want cfg3 && { prof cfg3_stats --stats -d $O/cfg3_stats -o r -- $B3;  grep '^{' $O/cfg3_stats.log | tail -1 > $O/r03_cfg3_bench_under_rocprof.json; }
want cfg5 && { prof cfg5_stats --stats -d $O/cfg5_stats -o r -- $B5;  grep '^{' $O/cfg5_stats.log | tail -1 > $O/r03_cfg5_bench_under_rocprof.json; }
# MFMA / traffic counters for the TIAGo and human shapes as well (one pass each)
want cfg3 && prof cfg3_mfma --pmc SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU -d $O/cfg3_mfma -o r -- $B3
want cfg3 && prof cfg3_fetch --pmc FETCH_SIZE -d $O/cfg3_fetch -o r -- $B3   # (FETCH_SIZE and WRITE_SIZE never in one pass: that
want cfg3 && prof cfg3_write --pmc WRITE_SIZE -d $O/cfg3_write -o r -- $B3   #  combination aborted the profiler and cost 40 GPU-minutes)
want cfg5 && prof cfg5_mfma --pmc SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU -d $O/cfg5_mfma -o r -- $B5
want cfg5 && prof cfg5_fetch --pmc FETCH_SIZE -d $O/cfg5_fetch -o r -- $B5
want cfg5 && prof cfg5_write --pmc WRITE_SIZE -d $O/cfg5_write -o r -- $B5
want cfg3 && python3 tools/pmc_summary.py $O/r03_pmc_summary_cfg3.json $O/cfg3_mfma/r_results.db $O/cfg3_fetch/r_results.db $O/cfg3_write/r_results.db > $O/r03_pmc_cfg3.txt 2>&1
want cfg5 && python3 tools/pmc_summary.py $O/r03_pmc_summary_cfg5.json $O/cfg5_mfma/r_results.db $O/cfg5_fetch/r_results.db $O/cfg5_write/r_results.db > $O/r03_pmc_cfg5.txt 2>&1
for c in $SEL; do
  python3 tools/rocpd_summary.py $O/${c}_stats/r_results.db > $O/r03_${c}_kernel_stats.txt 2>&1
done
want cfg2 && python3 tools/pmc_summary.py $O/r03_pmc_summary.json $O/cfg2_fetch/r_results.db $O/cfg2_write/r_results.db $O/cfg2_sq/r_results.db > $O/r03_pmc_cfg2.txt 2>&1
want cfg4 && python3 tools/pmc_summary.py $O/r03_pmc_summary_cfg4.json $O/cfg4_fetch/r_results.db $O/cfg4_write/r_results.db $O/cfg4_mfma/r_results.db $O/cfg4_mfma2/r_results.db > $O/r03_pmc_cfg4.txt 2>&1
find $O -name "*.db" -size +20M -delete
ls -la $O | head -40
