#!/bin/bash
# round-3 GPU call 11: launch occupancy over time (wall-clock timeline of the work items)
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r03k; mkdir -p $O; cd $R
for c in C1 C2 C4; do L3D_LIB=$R/gpurun_scratch/libl3dpp_hip_cyc.so timeout 300 python tools/cycles_run.py $c 2> $O/cyc_$c.err | tee $O/cyc_$c.json; tail -2 $O/cyc_$c.err; done
