# Where the waves of the Laplacian chain kernels spend their cycles (two rocprofv3 --pmc passes, <= 8 SQ counters each)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
tag=${1:-base}; envs=${2:-}
for kv in ${envs//,/ }; do export $kv; done
export REPS=5
rm -rf gpurun_out/pmc_${tag}_WAITA gpurun_out/pmc_${tag}_WAITB
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS --output-format csv -d gpurun_out/pmc_${tag}_WAITA -- python3 tools/spmm_micro.py > gpurun_out/pmc_${tag}_WAITA.log 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_IFETCH SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d gpurun_out/pmc_${tag}_WAITB -- python3 tools/spmm_micro.py > gpurun_out/pmc_${tag}_WAITB.log 2>&1 || exit 1
echo done
