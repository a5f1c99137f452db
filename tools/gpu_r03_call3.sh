#!/bin/bash
# round-3 GPU call 3: halo form / block cache / record-only tail through the test suite; what the match kernel's waves
# wait for (per-item timeline of the diagnostic build, PMC diagnostics); cold / second-scene times with the block cache
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r03c; mkdir -p $O; cd $R
( time timeout 1700 python -m pytest tests -m gpu -q ) > $O/pytest.log 2>&1; tail -8 $O/pytest.log
for c in C1 C4; do L3D_LIB=$R/gpurun_scratch/libl3dpp_hip_stats.so timeout 300 python tools/cycles_run.py $c > $O/cycles_$c.json 2> $O/cycles_$c.err; cat $O/cycles_$c.json; done
bash tools/pmc_diag.sh r03 C1 > $O/diag_c1.log 2>&1; cat gpurun_out/r03_pmc_diag_C1.txt
bash tools/pmc_diag.sh r03 C4 > $O/diag_c4.log 2>&1; cat gpurun_out/r03_pmc_diag_C4.txt
timeout 600 python bench.py > $O/bench_c1.json 2> $O/bench_c1.err; cut -c1-1200 $O/bench_c1.json
timeout 600 python bench.py --config C2 --no-cpu-baseline --steps 5 --warmup 2 > $O/bench_c2.json 2> $O/bench_c2.err; cut -c1-900 $O/bench_c2.json
