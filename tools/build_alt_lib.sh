#!/bin/bash
# A/B builds of the match kernel: tools/build_alt_lib.sh <name> <extra hipcc flags...>
#   -> gpurun_scratch/libl3dpp_hip_<name>.so = the product objects with k_match.hip recompiled with the flags
# (select with L3D_LIB=<path>; the build id is the product's: the variants are for timing, not for profiles)
set -e
name=$1; shift
R=$(cd "$(dirname "$0")/.." && pwd)
cd $R/line3dpp_amd/csrc
make -j8 > /dev/null
BID=$(cat l3d_api.hip l3d_affinity_host.hip l3d_access.hip l3d_output.hip l3d_seam.hip k_match.hip k_lists.hip k_views.hip k_scan.hip k_affinity.hip k_rdd.hip l3d_recon.hip l3d_neighbors.hip l3d_io.hip k_selftest.hip l3d_*.h ../../include/l3dpp_hip.h | md5sum | cut -c1-12)
mkdir -p $R/gpurun_scratch/alt_$name
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -fno-slp-vectorize "$@" -DL3D_BUILD_ID=\"$BID\" -c k_match.hip -o $R/gpurun_scratch/alt_$name/k_match.o
objs=$(ls *.o | grep -v '^k_match.o$')
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/gpurun_scratch/libl3dpp_hip_$name.so $objs $R/gpurun_scratch/alt_$name/k_match.o
echo built $R/gpurun_scratch/libl3dpp_hip_$name.so "($*)"
