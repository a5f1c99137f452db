#!/bin/bash
# round-3 GPU call 6: scalar-unit economy of the match kernel's walk (s_bitset0, 32-bit load offsets, F and the camera
# centres out of the walk's live ranges, no SLP packing) -- tests, then A/B of the remaining choices
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r03f; mkdir -p $O; cd $R
( time timeout 1700 python -m pytest tests -m gpu -q ) > $O/pytest.log 2>&1; tail -6 $O/pytest.log
for c in C1 C2 C4; do for v in "" rcp lf w7 w6 slp; do
  L=""; [ -n "$v" ] && L="L3D_LIB=$R/gpurun_scratch/libl3dpp_hip_$v.so"
  env $L timeout 300 python bench.py --config $c --no-cpu-baseline --no-cold --steps 8 --warmup 2 2> $O/ab.err | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$c', '$v' or 'default', 'ms/step', d['ms_per_step'], 'kernel', d['roofline']['kernel_ms'], d['phase_ms'])"
done; done | tee $O/ab.txt
