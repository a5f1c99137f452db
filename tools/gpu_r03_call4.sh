#!/bin/bash
# round-3 GPU call 4: full test suite on the fixed build (record-only tail, halo form, block cache); A/B of how the target
# records reach the lanes of the match kernel (scalar loads / v_readlane from a per-chunk vector load) at 6 and 7 waves per SIMD
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r03d; mkdir -p $O; cd $R
( time timeout 1700 python -m pytest tests -m gpu -q ) > $O/pytest.log 2>&1; tail -6 $O/pytest.log
for c in C1 C2 C4; do for v in "" sl6 rl6 rl7; do
  L=""; [ -n "$v" ] && L="L3D_LIB=$R/gpurun_scratch/libl3dpp_hip_$v.so"
  env $L timeout 300 python bench.py --config $c --no-cpu-baseline --no-cold --steps 6 --warmup 2 2> $O/ab.err | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$c', '$v' or 'default(sl7)', 'ms/step', d['ms_per_step'], 'kernel', d['roofline']['kernel_ms'], d['phase_ms'])"
done; done | tee $O/ab.txt
timeout 600 python bench.py > $O/bench_c1.json 2> $O/bench_c1.err; cut -c1-1500 $O/bench_c1.json; tail -3 $O/bench_c1.err
