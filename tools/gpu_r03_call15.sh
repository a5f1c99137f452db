#!/bin/bash
# round-3 GPU call 15: ablation of call 14's changes
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r03o; mkdir -p $O; cd $R
for rnd in 1 2; do for c in C1 C2 C4; do for v in "" selb poss flat flatep old; do
  L=""; [ -n "$v" ] && L="L3D_LIB=$R/gpurun_scratch/libl3dpp_hip_$v.so"
  env $L timeout 300 python bench.py --config $c --no-cpu-baseline --no-cold --steps 8 --warmup 2 2> $O/ab.err | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$rnd $c', '$v' or 'default', 'ms/step', d['ms_per_step'], 'kernel', d['roofline']['kernel_ms'])"
done; done; done | tee $O/ab.txt
