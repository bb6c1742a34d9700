# closing measurements, part A: all GPU tests, the smoke entry point, the default bench plain and under rocprofv3, the other workloads, a two-rank rehearsal
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 500 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/r3f_full_gpu.log 2>&1 || { tail -n 20 gpurun_out/r3f_full_gpu.log; exit 1; }
tail -n 1 gpurun_out/r3f_full_gpu.log
timeout -k 10 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -n 1 || exit 1
timeout -k 10 300 python bench.py > gpurun_out/r3f_bench_default.json 2> gpurun_out/r3f_bench_default.err || exit 1
bash tools/run_bench_prof.sh 35 > gpurun_out/r3f_bprof_out.log 2>&1 || exit 1
cp gpurun_out/bench_kernel_stats.csv gpurun_out/r3f_bench_kernel_stats.csv
cp gpurun_out/bench_prof.log gpurun_out/r3f_bench_under_rocprof.log
for w in pubmed cora reddit; do
  timeout -k 10 300 python bench.py --workload $w --no-cpu-baseline > gpurun_out/r3f_bench_$w.json 2> gpurun_out/r3f_bench_$w.err || exit 1
done
rm -rf gpurun_out/bench_prof_reddit
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/bench_prof_reddit -- python3 bench.py --workload reddit --steps 30 --warmup 5 --no-cpu-baseline > gpurun_out/r3f_bench_reddit_prof.log 2>&1 || exit 1
cp $(ls gpurun_out/bench_prof_reddit/*/*kernel_stats.csv | tail -1) gpurun_out/r3f_bench_kernel_stats_reddit.csv
SPARC_BENCH_BACKEND=gloo SPARC_BENCH_DEVICE=0 timeout -k 10 400 python bench.py --gpus 2 --steps 30 --warmup 5 --batches-per-rank 6 > gpurun_out/r3f_bench_2ranks_one_gpu_gloo.json 2> gpurun_out/r3f_bench_2ranks.err || exit 1
python - <<'PY'
import json
for n in ("default", "pubmed", "cora", "reddit", "2ranks_one_gpu_gloo"):
    d = json.loads(open(f"gpurun_out/r3f_bench_{n}.json").read().strip().splitlines()[-1])
    print(n, d["n_gpus"], round(d["value"]), round(d["ms_per_step"], 3), round(d["roofline"]["frac"], 3), round(d["roofline"]["avg_kernel_us"], 2))
PY
