#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r04o; mkdir -p $O; cd $R
for rnd in 1 2 3; do for c in C1 C2 C4; do for v in "" prev; do
  L=""; [ -n "$v" ] && L="L3D_LIB=$R/gpurun_scratch/libl3dpp_hip_$v.so"
  env $L timeout 300 python bench.py --config $c --no-cpu-baseline --no-cold --steps 8 --warmup 2 2> $O/ab.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$rnd $c', '$v' or 'default', 'ms/step', d['ms_per_step'], 'kernel', d['roofline']['kernel_ms'])"
done; done; done | tee $O/ab.txt
