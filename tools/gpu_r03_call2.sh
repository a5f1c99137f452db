#!/bin/bash
# round-3 GPU call 2: the new match kernel (two-stage pipeline, walk-order copies, cheaper pre-filter, division-free
# orientation decision, unscaled IEEE div / sqrt) -- full test suite, A/B against the round-2 paths, PMC + fit, full-size
# parity of C2 / C4 through the stored reference records, kernel statistics
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r03b; mkdir -p $O; cd $R
( time timeout 1700 python -m pytest tests -m gpu -q ) > $O/pytest.log 2>&1; tail -5 $O/pytest.log
bash tools/valu_calib.sh r03 > $O/valu_calib.log 2>&1; tail -62 $O/valu_calib.log | cut -c1-150
# A/B on the kernel time (no CPU baseline, no cold call): default | single-stage | compiler's div/sqrt
for c in C1 C2 C4; do for v in "" "L3D_MATCH_STAGED=0" "L3D_NO_FASTMATH=1"; do
  env $v timeout 300 python bench.py --config $c --no-cpu-baseline --no-cold --steps 6 --warmup 2 2> $O/ab.err | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$c', '$v' or 'default', 'ms/step', d['ms_per_step'], 'kernel', d['roofline']['kernel_ms'], d['phase_ms'])"
done; done | tee $O/ab.txt
bash tools/pmc_bench.sh r03 C1 > $O/pmc.log 2>&1; tail -2 $O/pmc.log
bash tools/valu_fit.sh r03 > $O/fit.log 2>&1; tail -2 $O/fit.log
cp gpurun_out/r03_pmc_match.json gpurun_out/r03_valu_calibration.json gpurun_out/r03_valu_fit.json profiles/ 2>/dev/null
timeout 600 python bench.py > $O/bench_c1.json 2> $O/bench_c1.err; cat $O/bench_c1.json
for c in C2 C4; do
  timeout 900 python bench.py --config $c --parity-digest --steps 5 --warmup 2 > $O/bench_$c.json 2> $O/bench_$c.err; cut -c1-1800 $O/bench_$c.json
done
cd /tmp && export TMPDIR=/tmp
for c in C1 C2 C4; do
  timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_$c -o p -- python $R/bench.py --config $c --steps 5 --warmup 2 --no-cpu-baseline --no-cold > $O/prof_$c.log 2>&1
  python $R/tools/prof_summary.py $(find $O/prof_$c -name "*.db" | head -1) > $O/kernel_stats_$c.txt 2>&1; head -30 $O/kernel_stats_$c.txt
done
