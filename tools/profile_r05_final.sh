cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05b; mkdir -p $O
prof() { local name=$1; shift; timeout -k 10 300 rocprofv3 --kernel-trace "$@" > $O/$name.log 2>&1; }
prof cfg2_stats --stats -d $O/cfg2_stats -o r -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline
grep '^{' $O/cfg2_stats.log | tail -1 > $O/r05_cfg2_bench_under_rocprof.json
python3 tools/rocpd_summary.py $O/cfg2_stats/r_results.db > $O/r05_cfg2_kernel_stats.txt 2>&1
prof cfg4_stats --stats -d $O/cfg4_stats -o r -- python3 bench.py --config cfg4 --steps 3 --warmup 1
grep '^{' $O/cfg4_stats.log | tail -1 > $O/r05_cfg4_bench_under_rocprof.json
python3 tools/rocpd_summary.py $O/cfg4_stats/r_results.db > $O/r05_cfg4_kernel_stats.txt 2>&1
prof cfg5_stats --stats -d $O/cfg5_stats -o r -- python3 bench.py --config cfg5 --steps 2 --warmup 1
grep '^{' $O/cfg5_stats.log | tail -1 > $O/r05_cfg5_bench_under_rocprof.json
python3 tools/rocpd_summary.py $O/cfg5_stats/r_results.db > $O/r05_cfg5_kernel_stats.txt 2>&1
prof cfg3_stats --stats -d $O/cfg3_stats -o r -- python3 bench.py --config cfg3 --steps 3 --warmup 1
grep '^{' $O/cfg3_stats.log | tail -1 > $O/r05_cfg3_bench_under_rocprof.json
python3 tools/rocpd_summary.py $O/cfg3_stats/r_results.db > $O/r05_cfg3_kernel_stats.txt 2>&1
find $O -name "*.db" -size +20M -delete
