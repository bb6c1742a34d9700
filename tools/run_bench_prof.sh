# rocprofv3 kernel stats of the default bench run; summary copied to gpurun_out/bench_kernel_stats.csv
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/bench_prof
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/bench_prof -- python3 bench.py --steps ${1:-30} --warmup 5 --no-cpu-baseline > gpurun_out/bench_prof.log 2> gpurun_out/bench_prof.err || exit 1
cp $(ls gpurun_out/bench_prof/*/*kernel_stats.csv | tail -1) gpurun_out/bench_kernel_stats.csv
head -40 gpurun_out/bench_kernel_stats.csv | cut -c1-160
