#!/bin/bash
# quick validation of a feature branch on the GPU box: the parity / mode tests, then an on/off A/B of one environment switch
#   bash tools/gpu_branch_check.sh <tag> <ENV_SWITCH=value that turns the feature off>
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/${1:-branch}; mkdir -p $O; cd $R
( time python -m pytest tests/test_gpu_parity.py tests/test_gpu_modes.py -m gpu -q -x 2>&1 | tail -6 ) 2>&1 | tail -10
python -m pytest tests/test_gpu_at_size.py -m gpu -q -x -k "c1 or C1" 2>&1 | tail -3
show() { python - "$1" "$2" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[2], "ms/step", d["ms_per_step"], "kernel", d["roofline"].get("kernel_ms"), {k:v for k,v in d["phase_ms"].items() if k!="measured_in"}, "parity", (d.get("parity") or {}).get("ok"))
except Exception as e:
    print(sys.argv[2], "FAILED", e, open(sys.argv[1].replace(".json",".err")).read()[-600:])
PY
}
for round in 1 2; do
  env $2 python bench.py --config C1 --steps 40 --no-cpu-baseline --no-cold > $O/off_C1_$round.json 2> $O/off_C1_$round.err; show $O/off_C1_$round.json "off C1"
  python bench.py --config C1 --steps 40 --no-cpu-baseline --no-cold > $O/on_C1_$round.json 2> $O/on_C1_$round.err; show $O/on_C1_$round.json "on  C1"
done
env $2 python bench.py --config C0 --steps 40 --no-cpu-baseline --no-cold > $O/off_C0.json 2> $O/off_C0.err; show $O/off_C0.json "off C0"
python bench.py --config C0 --steps 40 --no-cpu-baseline --no-cold > $O/on_C0.json 2> $O/on_C0.err; show $O/on_C0.json "on  C0"
