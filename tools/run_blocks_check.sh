# the dense-block path after a change: its tests, the chain's randomised cross-check, the Reddit-shape bench plain and under rocprofv3
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 400 python -u -m pytest tests/test_gpu_blocks.py tests/test_gpu_parity.py -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/blocks_check_tests.log 2>&1 || { tail -n 20 gpurun_out/blocks_check_tests.log; exit 1; }
tail -n 1 gpurun_out/blocks_check_tests.log
timeout -k 10 300 python -u tests/fuzz_lap.py > gpurun_out/blocks_check_fuzz_lap.txt 2>&1 || { tail -n 20 gpurun_out/blocks_check_fuzz_lap.txt; exit 1; }
tail -n 2 gpurun_out/blocks_check_fuzz_lap.txt
timeout -k 10 300 python bench.py --workload reddit --no-cpu-baseline > gpurun_out/blocks_check_bench_reddit.json 2> gpurun_out/blocks_check_bench_reddit.err || exit 1
rm -rf gpurun_out/bench_prof_reddit
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/bench_prof_reddit -- python3 bench.py --workload reddit --steps 30 --warmup 5 --no-cpu-baseline > gpurun_out/blocks_check_reddit_prof.log 2>&1 || exit 1
cp $(ls gpurun_out/bench_prof_reddit/*/*kernel_stats.csv | tail -1) gpurun_out/blocks_check_kernel_stats_reddit.csv
python - <<'PY'
import json, csv
d = json.loads(open("gpurun_out/blocks_check_bench_reddit.json").read().strip().splitlines()[-1])
print("reddit", round(d["value"]), round(d["ms_per_step"], 3))
for r in list(csv.DictReader(open("gpurun_out/blocks_check_kernel_stats_reddit.csv")))[:14]:
    print(r["Name"][:70], r["Calls"], r["AverageNs"])
PY
