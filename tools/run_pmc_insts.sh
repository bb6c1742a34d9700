# Instruction mix of the Laplacian chain kernels (one rocprofv3 --pmc pass, 8 SQ counters)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
tag=${1:-base}; envs=${2:-}
for kv in ${envs//,/ }; do export $kv; done
export REPS=5
rm -rf gpurun_out/pmc_${tag}_INSTS
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAVES SQ_INSTS_BRANCH --output-format csv -d gpurun_out/pmc_${tag}_INSTS -- python3 tools/spmm_micro.py > gpurun_out/pmc_${tag}_INSTS.log 2>&1 || exit 1
echo done
