#!/bin/bash
# Round-5 profiles on the GPU box: kernel stats and PMC passes of the bench command (cfg2: the fused launch) and of the other
# configs.  Results under gpurun_out/r05/ (copied to profiles/ by hand).  rocprofv3 is given the program itself
# (python3 bench.py ...), counters in their own passes (never FETCH_SIZE and WRITE_SIZE in one pass).
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05; mkdir -p $O
SEL=${1:-"cfg2"}
want() { case " $SEL " in *" $1 "*) return 0;; *) return 1;; esac; }
prof() { # name, extra rocprof args..., -- command
  local name=$1; shift
  timeout -k 10 300 rocprofv3 --kernel-trace "$@" > $O/$name.log 2>&1
}
B2="python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline"
B2N="python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-fuse"
B3="python3 bench.py --config cfg3 --steps 3 --warmup 1"
B4="python3 bench.py --config cfg4 --steps 3 --warmup 1"
B5="python3 bench.py --config cfg5 --steps 2 --warmup 1"
if want cfg2; then
  prof cfg2_stats --stats -d $O/cfg2_stats -o r -- $B2;  grep '^{' $O/cfg2_stats.log | tail -1 > $O/r05_cfg2_bench_under_rocprof.json
  prof cfg2_fetch --pmc FETCH_SIZE -d $O/cfg2_fetch -o r -- $B2
  prof cfg2_write --pmc WRITE_SIZE -d $O/cfg2_write -o r -- $B2
  prof cfg2_sq --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_BUSY_CYCLES SQ_WAVES -d $O/cfg2_sq -o r -- $B2
  prof cfg2_sq2 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU_FMA_F64 -d $O/cfg2_sq2 -o r -- $B2
  python3 tools/rocpd_summary.py $O/cfg2_stats/r_results.db > $O/r05_cfg2_kernel_stats.txt 2>&1
  python3 tools/pmc_summary.py $O/r05_pmc_summary.json $O/cfg2_fetch/r_results.db $O/cfg2_write/r_results.db $O/cfg2_sq/r_results.db $O/cfg2_sq2/r_results.db > $O/r05_pmc_cfg2.txt 2>&1
  # the two-launch form on the same box, for the side-by-side
  prof cfg2n_stats --stats -d $O/cfg2n_stats -o r -- $B2N; grep '^{' $O/cfg2n_stats.log | tail -1 > $O/r05_cfg2_two_launch_bench_under_rocprof.json
  python3 tools/rocpd_summary.py $O/cfg2n_stats/r_results.db > $O/r05_cfg2_two_launch_kernel_stats.txt 2>&1
fi
for c in cfg3 cfg4 cfg5; do
  if want $c; then
    eval B=\$B${c#cfg}
    prof ${c}_stats --stats -d $O/${c}_stats -o r -- $B;  grep '^{' $O/${c}_stats.log | tail -1 > $O/r05_${c}_bench_under_rocprof.json
    prof ${c}_fetch --pmc FETCH_SIZE -d $O/${c}_fetch -o r -- $B
    prof ${c}_write --pmc WRITE_SIZE -d $O/${c}_write -o r -- $B
    prof ${c}_mfma --pmc SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU -d $O/${c}_mfma -o r -- $B
    python3 tools/rocpd_summary.py $O/${c}_stats/r_results.db > $O/r05_${c}_kernel_stats.txt 2>&1
    python3 tools/pmc_summary.py $O/r05_pmc_summary_${c}.json $O/${c}_fetch/r_results.db $O/${c}_write/r_results.db $O/${c}_mfma/r_results.db > $O/r05_pmc_${c}.txt 2>&1
  fi
done
if want cfg3a; then  # TIAGo, active joints (8 of 24 row blocks): kernel stats only
  prof cfg3a_stats --stats -d $O/cfg3a_stats -o r -- python3 bench.py --config cfg3 --active-joints --steps 3 --warmup 1 --no-cpu-baseline
  grep '^{' $O/cfg3a_stats.log | tail -1 > $O/r05_cfg3_active_bench_under_rocprof.json
  python3 tools/rocpd_summary.py $O/cfg3a_stats/r_results.db > $O/r05_cfg3_active_kernel_stats.txt 2>&1
fi
find $O -name "*.db" -size +20M -delete
ls -la $O | head -40
