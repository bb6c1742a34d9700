# usage: bash tools/run_micro.sh "0 1 2 8"   (SPARC_DBG values), results under gpurun_out/micro_<dbg>/
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/micro_*
for d in ${1:-0}; do
  SPARC_DBG=$d timeout -k 10 120 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/micro_$d -- python3 tools/spmm_micro.py > gpurun_out/micro_$d.log 2>&1 || exit 1
done
echo done
