#!/bin/bash
# same-box A/B of the tile form of k_match_pairs (round 5) against the row form:
#   bash tools/gpu_ab_tile.sh <tag> "<variants for C1>" "<variants for C2>" "<variants for C4>"
# a variant = <tile rows 0|16|32>[:<alt library name in gpurun_scratch>][@<L3D_MATCH_CLASSES 0|1>]   e.g. "0 0@0 16 32 16:tw7"
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/${1:-abt}; mkdir -p $O; cd $R
show() { python - "$1" "$2" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[2], "ms/step", d["ms_per_step"], "kernel", d["roofline"].get("kernel_ms"), {k:v for k,v in d["phase_ms"].items() if k!="measured_in"}, flush=True)
except Exception as e:
    print(sys.argv[2], "FAILED", e, open(sys.argv[1].replace(".json",".err")).read()[-400:], flush=True)
PY
}
run() { # config variant round steps
  local v=${2%%@*} cls=1; [[ "$2" == *@* ]] && cls=${2#*@}
  local tile=${v%%:*} alt=""; [[ "$v" == *:* ]] && alt=${v#*:}
  local lib="$R/line3dpp_amd/csrc/libl3dpp_hip.so"; [ -n "$alt" ] && lib="$R/gpurun_scratch/libl3dpp_hip_$alt.so"
  local f=$O/$1_$(echo $2 | tr ':@' '__')_$3
  L3D_MATCH_TILE=$tile L3D_MATCH_CLASSES=$cls L3D_LIB=$lib python bench.py --config $1 --steps $4 --no-cpu-baseline --no-cold > $f.json 2> $f.err
  show $f.json "$1 tile=$2"
}
for round in 1 2; do for v in $2; do run C1 $v $round 40; done; done
for v in $3; do run C2 $v 1 8; done
for v in $4; do run C4 $v 1 4; done
