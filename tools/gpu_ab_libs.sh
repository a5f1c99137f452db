#!/bin/bash
# same-box A/B of alternative builds of the library (gpurun_scratch/libl3dpp_hip_<name>.so, L3D_LIB) against the tree's:
#   bash tools/gpu_ab_libs.sh <tag> "<names for C1>" "<names for C2>"     ("tree" = the library in the tree)
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/${1:-abl}; mkdir -p $O; cd $R
if [ -z "$NO_TESTS" ]; then ( time python -m pytest tests -m gpu -q 2>&1 | tail -4 ) 2>&1 | tail -8; fi
show() { python - "$1" "$2" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[2], "ms/step", d["ms_per_step"], "kernel", d["roofline"].get("kernel_ms"), {k:v for k,v in d["phase_ms"].items() if k!="measured_in"})
except Exception as e:
    print(sys.argv[2], "FAILED", e, open(sys.argv[1].replace(".json",".err")).read()[-400:])
PY
}
run() { # config name round steps
  local lib="$R/line3dpp_amd/csrc/libl3dpp_hip.so"; [ "$2" != tree ] && lib="$R/gpurun_scratch/libl3dpp_hip_$2.so"
  L3D_LIB=$lib python bench.py --config $1 --steps $4 --no-cpu-baseline --no-cold > $O/$2_$1_$3.json 2> $O/$2_$1_$3.err; show $O/$2_$1_$3.json "$2 $1"
}
for round in 1 2; do for n in $2; do run C1 $n $round 40; done; done
for n in $3; do run C2 $n 1 8; done
