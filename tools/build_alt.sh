#!/bin/bash
# A/B builds: tools/build_alt.sh <name> <source.hip> <extra hipcc flags...>
#   -> gpurun_scratch/libl3dpp_hip_<name>.so = the product objects with <source.hip> recompiled with the flags
# (select with L3D_LIB=<path>; the build id is the product's: the variants are for timing, not for profiles)
set -e
name=$1; src=$2; shift; shift
R=$(cd "$(dirname "$0")/.." && pwd)
cd $R/line3dpp_amd/csrc
make -j8 > /dev/null
BID=$(make -pn 2>/dev/null | grep '^BUILD_ID :=' | awk '{print $3}')
mkdir -p $R/gpurun_scratch/alt_$name
extra=""; [ "$src" = k_match.hip ] && extra="-fno-slp-vectorize"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC $extra "$@" -DL3D_BUILD_ID=\"$BID\" -c $src -o $R/gpurun_scratch/alt_$name/obj.o
objs=$(ls *.o | grep -v "^${src%.hip}.o$")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/gpurun_scratch/libl3dpp_hip_$name.so $objs $R/gpurun_scratch/alt_$name/obj.o
echo built $R/gpurun_scratch/libl3dpp_hip_$name.so "($src $*)"
