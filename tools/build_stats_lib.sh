#!/bin/bash
# Diagnostic build of the library with the candidate / list counters compiled in (-DL3D_STATS; slow: atomics in the
# hot loops).  Output: gpurun_scratch/libl3dpp_hip_stats.so -- use with L3D_LIB=<that path> (line3dpp_amd/_lib.py).
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p $R/gpurun_scratch/stats_build
cd $R/line3dpp_amd/csrc
for f in l3d_api l3d_affinity_host l3d_access l3d_output l3d_seam k_match k_views k_affinity k_rdd l3d_recon; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -DL3D_STATS -c $f.hip -o $R/gpurun_scratch/stats_build/$f.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/gpurun_scratch/libl3dpp_hip_stats.so $R/gpurun_scratch/stats_build/*.o
echo built $R/gpurun_scratch/libl3dpp_hip_stats.so
