# HBM-side traffic and L2 hit rate of the Laplacian chain kernels: separate counter passes
# usage: bash tools/run_pmc.sh <tag> [ENV=..,ENV=..]     results under gpurun_out/pmc_<tag>_<pass>/
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
tag=${1:-base}; envs=${2:-}
for kv in ${envs//,/ }; do export $kv; done
export REPS=5
for pass in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES"; do
  name=$(echo $pass | cut -d' ' -f1)
  rm -rf gpurun_out/pmc_${tag}_$name
  timeout -s KILL 120 rocprofv3 --pmc $pass --output-format csv -d gpurun_out/pmc_${tag}_$name -- python3 tools/spmm_micro.py > gpurun_out/pmc_${tag}_$name.log 2>&1 || exit 1
done
echo done
