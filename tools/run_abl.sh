# Where the time of the windowed inner pass goes: the kernel with parts of its work switched off (developer switch
# SPARC_DBG, bits: 1 no slots, 2 no LDS gather, 4 no window staging, 8 / 16 / 32 no entry / own-row / row-factor
# loads, 128 no store; bits 8-128 need a library built with -DSPARC_ABLATE: bash tools/build_variant.sh ablate -DSPARC_ABLATE,
# then SPARC_HIP_LIB=sparc_amd/lib/variants/libsparc_ablate.so in every spec).  usage: bash tools/run_abl.sh ["tag:ENV=.. ..."]; results: gpurun_out/micro_<tag>
spec=${1:-"full: d1:SPARC_DBG=1 d2:SPARC_DBG=2 d58:SPARC_DBG=58 d128:SPARC_DBG=128 d56:SPARC_DBG=56 d184:SPARC_DBG=184 d186:SPARC_DBG=186"}
bash tools/run_micro.sh "$spec" || exit 1
cd $GRAFT_REPO_ROOT
python tools/read_micro.py "spmm_win<2, true, 1" "spmm_win<2, true, 4"
