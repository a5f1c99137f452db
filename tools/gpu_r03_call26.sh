#!/bin/bash
# round-3 GPU call 26: work-item timeline of the final build (cycles-only diagnostic build of the same sources)
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r03y; mkdir -p $O; cd $R
for c in C1 C2 C4; do L3D_LIB=$R/gpurun_scratch/libl3dpp_hip_cyc.so timeout 120 python tools/cycles_run.py $c 2> $O/cyc_$c.err | tee $O/cyc_$c.json | cut -c1-900; done
