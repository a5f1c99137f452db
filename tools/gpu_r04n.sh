#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r04n; mkdir -p $O; cd $R
( time timeout 1500 python -m pytest tests -m gpu -q -s ) > $O/pytest.log 2>&1; tail -5 $O/pytest.log; grep -h "^FAILED" $O/pytest.log
for c in C1 C2; do L3D_LIB=$R/gpurun_scratch/libl3dpp_hip_stats.so timeout 200 python tools/phase_a_stats.py $c 2>$O/stats_$c.err | tee $O/stats_$c.json | cut -c1-1200; done
for rnd in 1 2; do for c in C1 C2 C4; do for v in "" nodefer; do
  E=""; [ -n "$v" ] && E="L3D_NO_DEFER=1"
  env $E timeout 300 python bench.py --config $c --no-cpu-baseline --no-cold --steps 8 --warmup 2 2> $O/ab.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$rnd $c', '$v' or 'default', 'ms/step', d['ms_per_step'], 'kernel', d['roofline']['kernel_ms'])"
done; done; done | tee $O/ab.txt
