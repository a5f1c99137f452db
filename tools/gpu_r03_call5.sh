#!/bin/bash
# round-3 GPU call 5: 16-byte inverse records, pre-filter without reciprocals (A/B against the form with two v_rcp_f32
# and against the unpacked fp32 code), per-variant register budget; calibration incl. v_med3_f32; kernel statistics
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r03e; mkdir -p $O; cd $R
( time timeout 1700 python -m pytest tests -m gpu -q ) > $O/pytest.log 2>&1; tail -6 $O/pytest.log
for c in C1 C2 C4; do for v in "" rcp noslp; do
  L=""; [ -n "$v" ] && L="L3D_LIB=$R/gpurun_scratch/libl3dpp_hip_$v.so"
  env $L timeout 300 python bench.py --config $c --no-cpu-baseline --no-cold --steps 8 --warmup 2 2> $O/ab.err | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$c', '$v' or 'default(products)', 'ms/step', d['ms_per_step'], 'kernel', d['roofline']['kernel_ms'], d['phase_ms'])"
done; done | tee $O/ab.txt
bash tools/valu_calib.sh r03 > $O/valu_calib.log 2>&1; grep -E "med3|add_abs|min_f32|max_f32|\"fma_f32\"" -A3 gpurun_out/r03_valu_calibration.json | head -40
cd /tmp && export TMPDIR=/tmp
for c in C1 C2 C4; do
  timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_$c -o p -- python $R/bench.py --config $c --steps 5 --warmup 2 --no-cpu-baseline --no-cold > $O/prof_$c.log 2>&1
  python $R/tools/prof_summary.py $(find $O/prof_$c -name "*.db" | head -1) > $O/kernel_stats_$c.txt 2>&1; head -12 $O/kernel_stats_$c.txt
done
