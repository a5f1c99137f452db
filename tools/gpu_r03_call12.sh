#!/bin/bash
# round-3 GPU call 12: compact LDS layout of the match kernel (7 waves per SIMD for one-wave items too) -- tests, A/B
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r03l; mkdir -p $O; cd $R
( time timeout 1700 python -m pytest tests -m gpu -q ) > $O/pytest.log 2>&1; tail -6 $O/pytest.log
for rnd in 1 2; do for c in C1 C2 C4; do for v in "" lF w6 lFw6; do
  L=""; [ -n "$v" ] && L="L3D_LIB=$R/gpurun_scratch/libl3dpp_hip_$v.so"
  env $L timeout 300 python bench.py --config $c --no-cpu-baseline --no-cold --steps 8 --warmup 2 2> $O/ab.err | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$rnd $c', '$v' or 'default', 'ms/step', d['ms_per_step'], 'kernel', d['roofline']['kernel_ms'])"
done; done; done | tee $O/ab.txt
