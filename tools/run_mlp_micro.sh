# usage: bash tools/run_mlp_micro.sh "<tag>:<ENV=..,ENV=..> ..."   kernel durations of tools/mlp_kernels_micro.py
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for spec in ${1:-base:}; do
  tag=${spec%%:*}; envs=${spec#*:}
  rm -rf gpurun_out/mlpm_$tag
  ( for kv in ${envs//,/ }; do export $kv; done
    timeout -k 10 150 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/mlpm_$tag -- python3 tools/mlp_kernels_micro.py > gpurun_out/mlpm_$tag.log 2>&1 ) || exit 1
  python - <<PY
import csv, glob
f = sorted(glob.glob("gpurun_out/mlpm_$tag/*/*kernel_stats.csv"))[-1]
for r in csv.DictReader(open(f)):
    if "anonymous" in r["Name"] and int(r["Calls"]) >= 10:
        print("$tag", f"{r['Name'][:90]:90s} calls={r['Calls']:>4s} avg_us={float(r['AverageNs'])/1e3:8.2f} min={float(r['MinNs'])/1e3:.2f}")
PY
done
