#!/bin/bash
# usage (on the GPU box): tools/valu_fit.sh <tag> -> gpurun_out/<tag>_valu_fit.json (see tools/valu_fit_run.py)
tag=${1:-r03}
R=${GRAFT_REPO_ROOT:-$(pwd)}
out=$R/gpurun_out/valufit_$tag
mkdir -p $out
[ -f $R/gpurun_scratch/libl3dpp_hip_stats.so ] || bash $R/tools/build_stats_lib.sh > $out/stats_build.log 2>&1
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES --output-format csv -d $out/p -o p -- python $R/tools/valu_fit_run.py > $out/product.json 2> $out/product.err
cd $R
L3D_LIB=$R/gpurun_scratch/libl3dpp_hip_stats.so timeout 300 python tools/valu_fit_run.py > $out/stats.json 2> $out/stats.err
python tools/valu_fit.py $out $tag
