#!/bin/bash
# round-3 GPU call 1: test suite, VALU calibration, allocation costs, bench C1 (+ cold call), PMC incl. class counters,
# VALU fit, bench C1 again quoting them, full-size C2 / C4 against the stored reference records
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r03a; mkdir -p $O; cd $R
date > $O/start.txt
K=""; [ -f tests/golden/full/C4.json ] || K="-k not_c4"; K=${K/not_c4/not full_c4}
( time timeout 1500 python -m pytest tests -m gpu -x -q ${K:+-k "not full_c4"} ) > $O/pytest.log 2>&1
tail -3 $O/pytest.log
bash tools/valu_calib.sh r03 > $O/valu_calib.log 2>&1; tail -45 $O/valu_calib.log
timeout 120 $R/tools/bin/alloc_bench > $O/alloc.json 2> $O/alloc.err; cat $O/alloc.json
timeout 600 python bench.py > $O/bench_c1_a.json 2> $O/bench_c1_a.err; cut -c1-1500 $O/bench_c1_a.json
bash tools/pmc_bench.sh r03 C1 > $O/pmc.log 2>&1; tail -3 $O/pmc.log
bash tools/valu_fit.sh r03 > $O/fit.log 2>&1; tail -3 $O/fit.log
cp gpurun_out/r03_pmc_match.json gpurun_out/r03_valu_calibration.json gpurun_out/r03_valu_fit.json profiles/ 2>/dev/null
timeout 600 python bench.py > $O/bench_c1.json 2> $O/bench_c1.err; cat $O/bench_c1.json
for c in C2 C4; do
  D="--parity-digest"; [ -f tests/golden/full/$c.json ] || D=""
  timeout 900 python bench.py --config $c $D --steps 5 --warmup 2 > $O/bench_$c.json 2> $O/bench_$c.err; cut -c1-2500 $O/bench_$c.json
done
date > $O/end.txt
