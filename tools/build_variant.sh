# usage: bash tools/build_variant.sh NAME "-DSPARC_X=1 ..."   -> sparc_amd/lib/variants/libsparc_NAME.so
# (spmm.hip compiled with extra defines, linked with the other objects of the regular build; for SPARC_HIP_LIB=...)
set -e
cd "$(dirname "$0")/.."
python -m sparc_amd.build > /dev/null
mkdir -p sparc_amd/lib/variants
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-pass-failed $2 -c sparc_amd/csrc/spmm.hip -o sparc_amd/lib/variants/spmm_$1.o
objs=$(ls sparc_amd/lib/*.o | grep -v "/spmm.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o sparc_amd/lib/variants/libsparc_$1.so sparc_amd/lib/variants/spmm_$1.o $objs
echo sparc_amd/lib/variants/libsparc_$1.so
