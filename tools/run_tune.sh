# GEMM tuning probe: tune, then rocprofv3 kernel stats of default vs tuned picks.  usage: bash tools/run_tune.sh [M]
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
M=${1:-48979}
timeout -k 10 400 python3 tools/tune_gemms.py tune $M gpurun_out/tunableop_$M.csv > gpurun_out/tune_$M.log 2>&1 || exit 1
rm -rf gpurun_out/tune_default_$M gpurun_out/tune_tuned_$M
USE_TUNED=0 timeout -k 10 120 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/tune_default_$M -- python3 tools/tune_gemms.py run $M gpurun_out/tunableop_$M.csv >> gpurun_out/tune_$M.log 2>&1 || exit 1
USE_TUNED=1 timeout -k 10 120 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/tune_tuned_$M -- python3 tools/tune_gemms.py run $M gpurun_out/tunableop_$M.csv >> gpurun_out/tune_$M.log 2>&1 || exit 1
grep -v "^W2026\|^E2026\|amdgpu.ids" gpurun_out/tune_$M.log
for v in default tuned; do
  f=$(ls gpurun_out/tune_${v}_$M/*/*kernel_stats.csv | tail -1)
  echo "== $v: GEMM kernels (rocprofv3 kernel stats, 23 calls per shape)"
  python3 - "$f" <<'PY'
import csv, sys
tot = 0.0
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Name"]
    if any(t in n for t in ("Cijk", "gemm", "Gemm", "GEMM")):
        print(f"   {n[:90]:90s} calls={r['Calls']:>4s} avg_us={float(r['AverageNs'])/1e3:9.1f}")
        tot += float(r["TotalDurationNs"])
print(f"   total GEMM time {tot/1e3/23:.1f} us per pass over the 7 shapes")
PY
done
