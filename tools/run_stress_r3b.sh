# degree-ordered R-MAT with long rows in chunks: row limits, then the HBM-side counters
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 400 python tools/spmm_stress.py rmat-degree-split=2048 rmat-degree-split=256 rmat-degree-split=128 > gpurun_out/r3_stress_b.jsonl 2> gpurun_out/r3_stress_b.err || exit 1
bash tools/run_stress_pmc.sh "rmat-degree-split=512" > gpurun_out/r3_stress_pmc_degree.txt 2>&1 || exit 1
cut -c1-420 gpurun_out/r3_stress_b.jsonl; tail -4 gpurun_out/r3_stress_pmc_degree.txt
