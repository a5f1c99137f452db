#!/bin/bash
# everything a GPU call runs must be built from the CURRENT sources before the snapshot is taken: product library,
# diagnostic (-DL3D_STATS) library, tool binaries, checkers
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
make -C $R/line3dpp_amd/csrc -j8 2>&1 | grep -E "error|Error" && exit 1
bash $R/tools/build_stats_lib.sh | tail -1
mkdir -p $R/tools/bin
for t in valu_calib alloc_bench; do
  [ $R/tools/bin/$t -nt $R/tools/$t.hip ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -o $R/tools/bin/$t $R/tools/$t.hip
done
rm -f $R/tools/bin/*.hipfb
make -C $R/oracle -s
python - <<PY
import sys; sys.path.insert(0, "$R")
from line3dpp_amd import _lib
import ctypes as C, os
a = _lib.load().l3d_build_info().decode()
b = C.CDLL(os.path.join("$R", "gpurun_scratch", "libl3dpp_hip_stats.so")); b.l3d_build_info.restype = C.c_char_p
print("product:", a); print("stats  :", b.l3d_build_info().decode())
assert a == b.l3d_build_info().decode(), "stats library is of another build"
PY
