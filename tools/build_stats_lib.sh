#!/bin/bash
# Diagnostic build of the library with the candidate / list counters compiled in (-DL3D_STATS; slow: atomics in the
# hot loops).  Output: gpurun_scratch/libl3dpp_hip_stats.so -- use with L3D_LIB=<that path> (line3dpp_amd/_lib.py).
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p $R/gpurun_scratch/stats_build
cd $R/line3dpp_amd/csrc
BID=$(cat l3d_api.hip l3d_affinity_host.hip l3d_access.hip l3d_output.hip l3d_seam.hip k_match.hip k_lists.hip k_views.hip k_scan.hip k_affinity.hip k_rdd.hip l3d_recon.hip l3d_neighbors.hip l3d_io.hip k_selftest.hip l3d_*.h ../../include/l3dpp_hip.h | md5sum | cut -c1-12)
for f in l3d_api l3d_affinity_host l3d_access l3d_output l3d_seam k_match k_lists k_views k_scan k_affinity k_rdd l3d_recon l3d_neighbors l3d_io k_selftest; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC $([ $f = k_match ] && echo -fno-slp-vectorize) -DL3D_STATS -DL3D_BUILD_ID=\"$BID\" -c $f.hip -o $R/gpurun_scratch/stats_build/$f.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/gpurun_scratch/libl3dpp_hip_stats.so $R/gpurun_scratch/stats_build/*.o
echo built $R/gpurun_scratch/libl3dpp_hip_stats.so
