# HBM-side bytes of the whole-graph products (rocprofv3 --pmc, separate passes)   usage: bash tools/run_stress_pmc.sh "<cases>"
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for pass in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum"; do
  name=$(echo $pass | cut -d' ' -f1)
  rm -rf gpurun_out/stress_pmc_$name
  timeout -s KILL 300 rocprofv3 --pmc $pass --output-format csv -d gpurun_out/stress_pmc_$name -- python3 tools/spmm_stress.py ${1:-rmat} > gpurun_out/stress_pmc_$name.log 2>&1 || exit 1
done
python3 - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/stress_pmc_*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if any(t in r["Kernel_Name"] for t in ("spmm_rows", "spmm_win", "spmm_chunks", "spmm_merge")):
            acc[r["Kernel_Name"][:50]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in acc.items():
    line = {c: sum(v) / len(v) for c, v in cs.items()}
    hbm = (2 * line.get("FETCH_SIZE", 0) + line.get("WRITE_SIZE", 0)) * 1024
    print(k, {c: round(v) for c, v in line.items()}, "HBM bytes per launch (2*FETCH+WRITE)*1024 =", round(hbm / 1e9, 2), "GB")
PY
