# usage: bash tools/run_blocks_micro.sh "<tag>:<ENV=..,..> ..."   results under gpurun_out/bmicro_<tag>/
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for spec in ${1:-base:}; do
  tag=${spec%%:*}; envs=${spec#*:}
  rm -rf gpurun_out/bmicro_$tag
  ( for kv in ${envs//,/ }; do export $kv; done
    timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/bmicro_$tag -- python3 tools/blocks_micro.py > gpurun_out/bmicro_$tag.log 2>&1 ) || exit 1
done
echo done
