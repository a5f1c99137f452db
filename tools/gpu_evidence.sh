#!/bin/bash
# The evidence of a round for the build in the tree, in two GPU calls (run `tools/build_stats_lib.sh` first):
#   gpurun --timeout 1500 -- 'bash tools/gpu_evidence.sh pmc r04'
#       PMC summaries of C1 / C2 / C4 / C3 / C0, the VALU fit, the wait / cache diagnostics of C1 and C4.  Afterwards copy
#       gpurun_out/<tag>*_pmc_match.json, <tag>_valu_fit.json and <tag>_v2_pmc_diag_*.txt into profiles/: bench.py
#       quotes a summary only when its build id is the id of the library it times.
#   gpurun --timeout 1500 -- 'bash tools/gpu_evidence.sh bench r04'
#       smoke, the five bench lines (C2 / C4 against the stored full-size records), kernel statistics of C1 / C2 / C4
#       -> gpurun_out/<tag>final/
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; T=${2:-r06}
if [ "$1" = pmc ]; then
  ( time bash tools/pmc_bench.sh $T C1 ) 2>&1 | tail -3
  ( time bash tools/valu_fit.sh $T ) 2>&1 | tail -2
  ( time bash tools/pmc_diag.sh ${T}_v2 C1 ) 2>&1 | tail -4 | cut -c1-200
  for c in 2 4 3 0; do ( time bash tools/pmc_bench.sh ${T}c$c C$c ) 2>&1 | tail -3; done
  ( time bash tools/pmc_diag.sh ${T}_v2 C4 ) 2>&1 | tail -4 | cut -c1-200
  exit 0
fi
O=$R/gpurun_out/${T}final; mkdir -p $O
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | cut -c1-200
( time python bench.py ) > $O/bench_C1.json 2> $O/bench_C1.err; tail -c 400 $O/bench_C1.err
( time python bench.py --config C2 --parity-digest ) > $O/bench_C2.json 2> $O/bench_C2.err
( time python bench.py --config C4 --parity-digest ) > $O/bench_C4.json 2> $O/bench_C4.err
( time python bench.py --config C3 ) > $O/bench_C3.json 2> $O/bench_C3.err
( time python bench.py --config C0 ) > $O/bench_C0.json 2> $O/bench_C0.err
for c in C1 C2 C4 C3 C0; do python - $O/bench_$c.json $c <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    r=d["roofline"]; p=d.get("parity") or {}; cb=d.get("cpu_baseline") or {}
    print(sys.argv[2], "ms/step", d["ms_per_step"], "value", d["value"], "kernel", r.get("kernel_ms"), "roof", r.get("bound"), r.get("frac"), "useful", r.get("useful_frac"),
          "traffic/algo", (r.get("hbm") or {}).get("traffic_over_algorithmic"), "parity", p.get("ok"), p.get("floats_checked"), p.get("max_rel"), "cpu", cb.get("value"), cb.get("cores"), "cold", d.get("cold_ms"), (d.get("cold") or {}).get("fresh_process"))
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
done
for c in C1 C2 C4; do bash tools/prof_bench.sh ${T}final_$c --config $c --no-cold > $O/kernel_stats_$c.txt 2>&1; head -12 $O/kernel_stats_$c.txt; done
