#!/bin/bash
# round-3 GPU call 16: the bench lines of the final build with the corrected PMC summaries of C2 / C4 and the direct
# record exchange in the N-GPU model
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r03p; mkdir -p $O; cd $R
timeout 900 python bench.py > $O/bench_C1.json 2> $O/bench_C1.err; cut -c1-300 $O/bench_C1.json; tail -2 $O/bench_C1.err
for c in C2 C4; do
  timeout 900 python bench.py --config $c --parity-digest --steps 5 --warmup 2 > $O/bench_$c.json 2> $O/bench_$c.err; cut -c1-300 $O/bench_$c.json
done
for c in C0 C3; do
  timeout 900 python bench.py --config $c --steps 10 --warmup 2 > $O/bench_$c.json 2> $O/bench_$c.err; cut -c1-300 $O/bench_$c.json
done
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -3 $O/smoke.log
