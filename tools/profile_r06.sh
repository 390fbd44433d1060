#!/bin/bash
# Round-6 profiles on the GPU box -- ONE script, ONE output tree (gpurun_out/r06/prof), every summary stamped with the library
# it was taken on (round 5 mixed three trees).  Usage, from the build container:
#   gpurun --timeout 1500 -- "PROF_GIT_HEAD=$(git rev-parse --short HEAD) bash tools/profile_r06.sh 'cfg2 cfg3 cfg4 cfg5'"
# rocprofv3 is given the program itself (python3 bench.py ...); counters in their own passes (never FETCH_SIZE and WRITE_SIZE
# in one pass, never --pmc together with a trace domain other than --kernel-trace).
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06/prof; mkdir -p $O
SEL=${1:-"cfg2"}
STAMP="# library: libfigh.so sha256 $(sha256sum figaroh_plus_amd/libfigh.so | cut -c1-16), source tree at commit ${PROF_GIT_HEAD:-unknown} (+ uncommitted changes if the hash of a later build differs); $(date -u +%Y-%m-%dT%H:%MZ); tools/profile_r06.sh"
want() { case " $SEL " in *" $1 "*) return 0;; *) return 1;; esac; }
prof() { local name=$1; shift; timeout -k 10 400 rocprofv3 --kernel-trace "$@" > $O/$name.log 2>&1; }
stamp() { local f=$1; { echo "$STAMP"; cat $f; } > $f.tmp && mv $f.tmp $f; }
COMMON="--no-cpu-baseline --strong-config="
B2="python3 bench.py --steps 20 --warmup 5 $COMMON"
B2N="python3 bench.py --steps 10 --warmup 3 --no-fuse $COMMON"
B3="python3 bench.py --config cfg3 --steps 5 --warmup 2 $COMMON"
B4="python3 bench.py --config cfg4 --steps 3 --warmup 1 $COMMON"
B5="python3 bench.py --config cfg5 --steps 2 --warmup 1 $COMMON"
for c in cfg2 cfg3 cfg4 cfg5; do
  if want $c; then
    eval B=\$B${c#cfg}
    prof ${c}_stats --stats -d $O/${c}_stats -o r -- $B;  grep '^{' $O/${c}_stats.log | tail -1 > $O/r06_${c}_bench_under_rocprof.json
    python3 tools/rocpd_summary.py $O/${c}_stats/r_results.db > $O/r06_${c}_kernel_stats.txt 2>&1; stamp $O/r06_${c}_kernel_stats.txt
    if [ $c = cfg3 ]; then python3 tools/rocpd_timeline.py $O/cfg3_stats/r_results.db "regressor_tape_kernel<16" > $O/r06_cfg3_timeline.txt 2>&1; stamp $O/r06_cfg3_timeline.txt; fi
    prof ${c}_fetch --pmc FETCH_SIZE -d $O/${c}_fetch -o r -- $B
    prof ${c}_write --pmc WRITE_SIZE -d $O/${c}_write -o r -- $B
    if [ $c = cfg2 ]; then
      prof cfg2_sq --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_BUSY_CYCLES SQ_WAVES -d $O/cfg2_sq -o r -- $B
      prof cfg2_sq2 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU_FMA_F64 -d $O/cfg2_sq2 -o r -- $B
      python3 tools/pmc_summary.py $O/r06_pmc_summary.json $O/cfg2_fetch/r_results.db $O/cfg2_write/r_results.db $O/cfg2_sq/r_results.db $O/cfg2_sq2/r_results.db > $O/r06_pmc_cfg2.txt 2>&1
      stamp $O/r06_pmc_cfg2.txt
    else
      prof ${c}_mfma --pmc SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU -d $O/${c}_mfma -o r -- $B
      python3 tools/pmc_summary.py $O/r06_pmc_summary_${c}.json $O/${c}_fetch/r_results.db $O/${c}_write/r_results.db $O/${c}_mfma/r_results.db > $O/r06_pmc_${c}.txt 2>&1
      stamp $O/r06_pmc_${c}.txt
    fi
  fi
done
if want cfg2n; then  # the two-launch form on the same box, for the side-by-side
  prof cfg2n_stats --stats -d $O/cfg2n_stats -o r -- $B2N; grep '^{' $O/cfg2n_stats.log | tail -1 > $O/r06_cfg2_two_launch_bench_under_rocprof.json
  python3 tools/rocpd_summary.py $O/cfg2n_stats/r_results.db > $O/r06_cfg2_two_launch_kernel_stats.txt 2>&1; stamp $O/r06_cfg2_two_launch_kernel_stats.txt
fi
if want cfg3a; then  # TIAGo, active joints (8 of 24 row blocks): kernel stats only
  prof cfg3a_stats --stats -d $O/cfg3a_stats -o r -- python3 bench.py --config cfg3 --active-joints --steps 3 --warmup 1 $COMMON
  grep '^{' $O/cfg3a_stats.log | tail -1 > $O/r06_cfg3_active_bench_under_rocprof.json
  python3 tools/rocpd_summary.py $O/cfg3a_stats/r_results.db > $O/r06_cfg3_active_kernel_stats.txt 2>&1; stamp $O/r06_cfg3_active_kernel_stats.txt
fi
echo "$STAMP" > $O/r06_profile_stamp.txt
find $O -name "*.db" -size +20M -delete
ls -la $O | head -60
