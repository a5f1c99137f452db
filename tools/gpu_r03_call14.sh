#!/bin/bash
# round-3 GPU call 14: global (not flat) loads of the per-view records, branch-free selection in mutual_overlap
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r03n; mkdir -p $O; cd $R
( time timeout 1700 python -m pytest tests -m gpu -q -x ) > $O/pytest.log 2>&1; tail -4 $O/pytest.log
for rnd in 1 2; do for c in C1 C2 C4; do for v in ""; do
  env timeout 300 python bench.py --config $c --no-cpu-baseline --no-cold --steps 8 --warmup 2 2> $O/ab.err | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$rnd $c', '$v' or 'default', 'ms/step', d['ms_per_step'], 'kernel', d['roofline']['kernel_ms'], d['phase_ms'])"
done; done; done | tee $O/ab.txt
