#!/bin/bash
# same-box A/B of environment switches / alternative builds:
#   bash tools/gpu_ab_env.sh <tag> <config> <steps> <rounds> "<name>|<ENV=val ...>|<alt lib name or empty>" ...
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/$1; mkdir -p $O; cd $R
cfg=$2; steps=$3; rounds=$4; shift 4
show() { python - "$1" "$2" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[2], "ms/step", d["ms_per_step"], "kernel", d["roofline"].get("kernel_ms"), {k:v for k,v in d["phase_ms"].items() if k!="measured_in"}, flush=True)
except Exception as e:
    print(sys.argv[2], "FAILED", e, open(sys.argv[1].replace(".json",".err")).read()[-400:], flush=True)
PY
}
for round in $(seq 1 $rounds); do
  for v in "$@"; do
    IFS='|' read -r name envs alt <<< "$v"
    lib="$R/line3dpp_amd/csrc/libl3dpp_hip.so"; [ -n "$alt" ] && lib="$R/gpurun_scratch/libl3dpp_hip_$alt.so"
    env $envs L3D_LIB=$lib python bench.py --config $cfg --steps $steps --no-cpu-baseline --no-cold > $O/${cfg}_${name}_$round.json 2> $O/${cfg}_${name}_$round.err
    show $O/${cfg}_${name}_$round.json "$cfg $name"
  done
done
