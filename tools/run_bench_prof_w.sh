# rocprofv3 kernel stats of a bench run of the named workload: bash tools/run_bench_prof_w.sh reddit [steps]
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
w=${1:-reddit}
rm -rf gpurun_out/bench_prof_$w
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/bench_prof_$w -- python3 bench.py --workload $w --steps ${2:-30} --warmup 5 --no-cpu-baseline > gpurun_out/bench_prof_$w.log 2> gpurun_out/bench_prof_$w.err || exit 1
cp $(ls gpurun_out/bench_prof_$w/*/*kernel_stats.csv | tail -1) gpurun_out/bench_kernel_stats_$w.csv
