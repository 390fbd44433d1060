cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05c; mkdir -p $O
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/cfg3_stats -o r -- python3 bench.py --config cfg3 --steps 5 --warmup 2 --no-cpu-baseline > $O/cfg3_stats.log 2>&1
grep '^{' $O/cfg3_stats.log | tail -1 > $O/r05_cfg3_bench_under_rocprof.json
python3 tools/rocpd_summary.py $O/cfg3_stats/r_results.db > $O/r05_cfg3_kernel_stats.txt 2>&1
python3 tools/rocpd_timeline.py $O/cfg3_stats/r_results.db "regressor_tape_kernel<16" > $O/r05_cfg3_timeline.txt 2>&1
find $O -name "*.db" -size +20M -delete
tail -5 $O/r05_cfg3_timeline.txt
