# LDS bank conflicts and kernel time of the windowed kernel with and without the parity order of the entry streams
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for par in 0 1; do
  export PARITY=$par
  rm -rf gpurun_out/micro_par$par gpurun_out/pmc_par${par}_LDS
  timeout -k 10 150 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/micro_par$par -- python3 tools/spmm_micro.py > gpurun_out/micro_par$par.log 2>&1 || exit 1
  REPS=5 timeout -s KILL 120 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_BUSY_CYCLES --output-format csv -d gpurun_out/pmc_par${par}_LDS -- python3 tools/spmm_micro.py > gpurun_out/pmc_par${par}_LDS.log 2>&1 || exit 1
done
echo done
