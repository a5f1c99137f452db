#!/bin/bash
# usage (on the GPU box): tools/gpu_battery.sh <tag> [tests|notests] [configs...]
# the standard battery of a GPU call: pytest -m gpu, one bench line per config (no CPU baseline, no cold legs), and the
# rocprofv3 kernel statistics of C1 -> gpurun_out/<tag>/
tag=$1; shift; tests=${1:-tests}; shift; cfgs=${@:-C1 C2 C4}
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/$tag; mkdir -p $O; cd $R
if [ "$tests" = tests ]; then ( time timeout 1500 python -m pytest tests -m gpu -q -s ) > $O/pytest.log 2>&1; tail -5 $O/pytest.log; fi
for c in $cfgs; do
  timeout 600 python bench.py --config $c --no-cpu-baseline --no-cold --steps 10 --warmup 3 2> $O/bench_$c.err > $O/bench_$c.json
  python - $O/bench_$c.json $c <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    t=d.get("timings_ms") or d.get("phases") or {}
    print(sys.argv[2], "ms/step", d["ms_per_step"], "kernel", d["roofline"].get("kernel_ms"), {k:v for k,v in d.items() if k in ("phase_ms","phases_ms")})
except Exception as e:
    print(sys.argv[2], "bench failed:", e); print(open(sys.argv[1].replace(".json",".err")).read()[-1500:])
PY
done
cd /tmp && export TMPDIR=/tmp
for c in C1; do
  timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_$c -o p -- python $R/bench.py --config $c --steps 5 --warmup 2 --no-cpu-baseline --no-cold > $O/prof_$c.log 2>&1
  cd $R; python tools/prof_summary.py $(find $O/prof_$c -name "*.db" | head -1) > $O/kernel_stats_$c.txt 2>&1; head -24 $O/kernel_stats_$c.txt; cd /tmp
done
