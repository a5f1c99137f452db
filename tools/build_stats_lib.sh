#!/bin/bash
# Diagnostic build of the library with the candidate / list counters compiled in (-DL3D_STATS; slow: atomics in the
# hot loops).  Output: gpurun_scratch/libl3dpp_hip_stats.so -- use with L3D_LIB=<that path> (line3dpp_amd/_lib.py).
# The source list and the build id are the Makefile's (the id of the product build: pmc_to_json.py compares them).
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p $R/gpurun_scratch/stats_build
cd $R/line3dpp_amd/csrc
SRCS=$(make -pn 2>/dev/null | grep '^SRCS = ' | cut -d= -f2)
BID=$(make -pn 2>/dev/null | grep '^BUILD_ID :=' | awk '{print $3}')
rm -f $R/gpurun_scratch/stats_build/*.o
for s in $SRCS; do
  f=${s%.hip}
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC $([ $f = k_match ] && echo -fno-slp-vectorize) -DL3D_STATS -DL3D_BUILD_ID=\"$BID\" -c $f.hip -o $R/gpurun_scratch/stats_build/$f.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/gpurun_scratch/libl3dpp_hip_stats.so $R/gpurun_scratch/stats_build/*.o
echo built $R/gpurun_scratch/libl3dpp_hip_stats.so build=$BID
