# usage: bash tools/run_micro.sh "<tag>:<ENV=..,ENV=..> ..."   results under gpurun_out/micro_<tag>/
# e.g. bash tools/run_micro.sh "v2:VEC=2 v4:VEC=4,THREADS=1024"
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for spec in ${1:-base:}; do
  tag=${spec%%:*}; envs=${spec#*:}
  rm -rf gpurun_out/micro_$tag
  ( for kv in ${envs//,/ }; do export $kv; done
    timeout -k 10 150 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/micro_$tag -- python3 tools/spmm_micro.py > gpurun_out/micro_$tag.log 2>&1 ) || exit 1
done
echo done
