#!/bin/bash
# same-box A/B of the tree against the previous commit exported to gpurun_scratch/prev (git archive HEAD + make there)
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/${1:-ab}; mkdir -p $O; cd $R
( time python -m pytest tests -m gpu -q 2>&1 | tail -4 ) 2>&1 | tail -8
show() { python - "$1" "$2" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[2], "ms/step", d["ms_per_step"], "kernel", d["roofline"].get("kernel_ms"), {k:v for k,v in d["phase_ms"].items() if k!="measured_in"})
except Exception as e:
    print(sys.argv[2], "FAILED", e, open(sys.argv[1].replace(".json",".err")).read()[-400:])
PY
}
for round in 1 2; do
  for c in C1 C2 C4; do
    [ $round = 2 ] && [ $c != C1 ] && continue
    st=8; [ $c = C1 ] && st=40
    python gpurun_scratch/prev/bench.py --config $c --steps $st --no-cpu-baseline --no-cold > $O/prev_${c}_$round.json 2> $O/prev_${c}_$round.err; show $O/prev_${c}_$round.json "prev $c"
    python bench.py --config $c --steps $st --no-cpu-baseline --no-cold > $O/new_${c}_$round.json 2> $O/new_${c}_$round.err; show $O/new_${c}_$round.json "new  $c"
  done
done
bash tools/prof_bench.sh ${1:-ab}_C1 --config C1 --no-cold > $O/kernel_stats_C1.txt 2>&1; head -16 $O/kernel_stats_C1.txt
