# closing check on the GPU box: all GPU tests, the smoke entry point, the profiled and the default bench
cd $GRAFT_REPO_ROOT
timeout -k 10 500 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/full_gpu.log 2>&1 || { tail -n 20 gpurun_out/full_gpu.log; exit 1; }
tail -n 1 gpurun_out/full_gpu.log
timeout -k 10 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -n 1 || exit 1
bash tools/run_bench_prof.sh 35 > gpurun_out/bprof_out.log 2>&1 || exit 1
timeout -k 10 300 python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err || exit 1
python - <<'PY'
import json
d = json.load(open("gpurun_out/bench_default.json"))
print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["avg_kernel_us"], d["config"]["prep_seconds_per_batch"])
PY
