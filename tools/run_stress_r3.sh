# config-5 products with long rows in chunks: timings, then the HBM-side counters of the split product
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 400 python tools/spmm_stress.py rmat rmat-split=2048 rmat-split=512 rmat-split=128 rmat-degree rmat-degree-split=512 > gpurun_out/r3_stress.jsonl 2> gpurun_out/r3_stress.err || exit 1
RMAT_NODES=1000000 RMAT_EDGES=25000000 timeout -k 10 200 python tools/spmm_stress.py rmat rmat-split=512 >> gpurun_out/r3_stress.jsonl 2>> gpurun_out/r3_stress.err || exit 1
bash tools/run_stress_pmc.sh "rmat-split=512" > gpurun_out/r3_stress_pmc.txt 2>&1 || exit 1
cut -c1-420 gpurun_out/r3_stress.jsonl; tail -4 gpurun_out/r3_stress_pmc.txt
