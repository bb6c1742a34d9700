# usage: bash tools/run_variant_bench.sh "base name1 name2"   the default bench under rocprofv3 for the in-tree library (base)
# and for variant libraries built by tools/build_variant.sh; results under gpurun_out/bp_<name>{.json,/}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for v in ${1:-base}; do
  rm -rf gpurun_out/bp_$v
  if [ $v != base ]; then export SPARC_HIP_LIB=$GRAFT_REPO_ROOT/sparc_amd/lib/variants/libsparc_$v.so; fi
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/bp_$v -- python3 bench.py --steps 35 --warmup 5 --no-cpu-baseline > gpurun_out/bp_$v.json 2> gpurun_out/bp_$v.err || exit 1
done
echo done
