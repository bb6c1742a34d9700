# HBM traffic of the Laplacian chain kernels: two separate counter passes (FETCH_SIZE, WRITE_SIZE)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/pmc_fetch gpurun_out/pmc_write
REPS=5 timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_fetch -- python3 tools/spmm_micro.py > gpurun_out/pmc_fetch.log 2>&1 || exit 1
REPS=5 timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_write -- python3 tools/spmm_micro.py > gpurun_out/pmc_write.log 2>&1 || exit 1
echo done
