#!/bin/bash
# round-3 GPU call 20: the judged artefacts of the final build -- tests, PMC passes (C1, C2, C4) + instruction fit,
# bench lines (C1 full; C2 / C4 with the full-size reference records), rocprofv3 kernel statistics
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r03t; mkdir -p $O; cd $R
( time timeout 1700 python -m pytest tests -m gpu -q ) > $O/pytest.log 2>&1; tail -5 $O/pytest.log
bash tools/pmc_bench.sh r03 C1 > $O/pmc_c1.log 2>&1; tail -2 $O/pmc_c1.log
bash tools/valu_fit.sh r03 > $O/fit.log 2>&1; tail -2 $O/fit.log
bash tools/pmc_bench.sh r03c2 C2 > $O/pmc_c2.log 2>&1; tail -1 $O/pmc_c2.log
bash tools/pmc_bench.sh r03c4 C4 > $O/pmc_c4.log 2>&1; tail -1 $O/pmc_c4.log
cp gpurun_out/r03_pmc_match.json gpurun_out/r03c2_pmc_match.json gpurun_out/r03c4_pmc_match.json gpurun_out/r03_valu_fit.json profiles/ 2>/dev/null
timeout 900 python bench.py > $O/bench_C1.json 2> $O/bench_C1.err; cut -c1-1200 $O/bench_C1.json; tail -3 $O/bench_C1.err
for c in C2 C4; do
  timeout 900 python bench.py --config $c --parity-digest --steps 5 --warmup 2 > $O/bench_$c.json 2> $O/bench_$c.err; cut -c1-600 $O/bench_$c.json
done
for c in C0 C3; do
  timeout 900 python bench.py --config $c --steps 10 --warmup 2 > $O/bench_$c.json 2> $O/bench_$c.err; cut -c1-400 $O/bench_$c.json
done
cd /tmp && export TMPDIR=/tmp
for c in C1 C2 C4; do
  timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_$c -o p -- python $R/bench.py --config $c --steps 5 --warmup 2 --no-cpu-baseline --no-cold > $O/prof_$c.log 2>&1
  python $R/tools/prof_summary.py $(find $O/prof_$c -name "*.db" | head -1) > $O/kernel_stats_$c.txt 2>&1; head -8 $O/kernel_stats_$c.txt
done
cd $R; python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -2 $O/smoke.log
