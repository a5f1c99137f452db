#!/bin/bash
# round-3 GPU call 10: why stage 2 rejects a third of its candidates; launch occupancy over time
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r03j; mkdir -p $O; cd $R
for c in C1 C2 C4; do L3D_LIB=$R/gpurun_scratch/libl3dpp_hip_stats.so timeout 300 python tools/phase_a_stats.py $c 2> $O/st_$c.err | tee $O/st_$c.json; done
for c in C1 C2 C4; do L3D_LIB=$R/gpurun_scratch/libl3dpp_hip_cyc.so timeout 300 python tools/cycles_run.py $c 2> $O/cyc_$c.err | tee $O/cyc_$c.json; done
