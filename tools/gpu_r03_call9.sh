#!/bin/bash
# round-3 GPU call 9: per-work-item timeline of the match kernel (cycles-only diagnostic build)
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r03i; mkdir -p $O; cd $R
for c in C1 C2 C4; do L3D_LIB=$R/gpurun_scratch/libl3dpp_hip_cyc.so timeout 300 python tools/cycles_run.py $c 2> $O/cyc_$c.err | tee $O/cyc_$c.json; done
for w in 1 2; do L3D_MATCH_WPG=$w timeout 300 python bench.py --config C1 --no-cpu-baseline --no-cold --steps 8 --warmup 2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('C1 WPG=$w', 'ms/step', d['ms_per_step'], 'kernel', d['roofline']['kernel_ms'])"; done
