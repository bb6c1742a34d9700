# closing measurements, part B: HBM-side counters of the inner pass, every randomised cross-check
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
bash tools/run_pmc.sh r3 > gpurun_out/r3f_pmc.log 2>&1 || exit 1
python tools/collect_traffic.py r3 "spmm_win<2, true, 1, 2>" > gpurun_out/r3f_traffic.json || exit 1
bash tools/run_all_fuzz.sh 2>&1 | tee gpurun_out/r3f_fuzz_all.txt
