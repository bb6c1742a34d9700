# usage: bash tools/build_variant.sh NAME "-DSPARC_X=1 ..." [file.hip]   -> sparc_amd/lib/variants/libsparc_NAME.so
# (one source, default spmm.hip, compiled with extra defines and linked with the other objects of the regular build;
# load it with SPARC_HIP_LIB=...)
set -e
cd "$(dirname "$0")/.."
python -m sparc_amd.build > /dev/null
src=${3:-spmm.hip}; base=${src%.hip}
mkdir -p sparc_amd/lib/variants
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-pass-failed $2 -c sparc_amd/csrc/$src -o sparc_amd/lib/variants/${base}_$1.o
objs=$(ls sparc_amd/lib/*.o | grep -v "/$base.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o sparc_amd/lib/variants/libsparc_$1.so sparc_amd/lib/variants/${base}_$1.o $objs
echo sparc_amd/lib/variants/libsparc_$1.so
