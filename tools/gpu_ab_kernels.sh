#!/bin/bash
# same-box A/B of library builds by KERNEL time: rocprofv3 --kernel-trace --stats of `bench.py --steps 5 --warmup 2` per variant
#   bash tools/gpu_ab_kernels.sh <tag> "<lib names: tree | name of gpurun_scratch/libl3dpp_hip_<name>.so>" "<configs>" "<kernel regex>" [env assignments]
R=${GRAFT_REPO_ROOT:-$(pwd)}; tag=$1; names=$2; cfgs=${3:-C1}; rex=${4:-k_}; shift 4
O=$R/gpurun_out/abk_$tag; mkdir -p $O
for c in $cfgs; do for n in $names; do
  lib="$R/line3dpp_amd/csrc/libl3dpp_hip.so"; [ "$n" != tree ] && lib="$R/gpurun_scratch/libl3dpp_hip_$n.so"
  cd /tmp && export TMPDIR=/tmp
  rm -rf $O/$n_$c
  env L3D_LIB=$lib "$@" timeout 300 rocprofv3 --kernel-trace --stats -d $O/${n}_$c -o t -- python $R/bench.py --config $c --steps 5 --warmup 2 --no-cpu-baseline --no-cold > $O/${n}_$c.json 2> $O/${n}_$c.err
  cd $R
  echo "== $n $c $(python -c "import json,sys; d=json.loads([l for l in open('$O/${n}_$c.json') if l.startswith('{')][-1]); print('ms/step', d['ms_per_step'], {k: v for k, v in d['phase_ms'].items() if k != 'measured_in'})" 2>&1 | tail -1)"
  python tools/prof_summary.py $(find $O/${n}_$c -name "*.db" | head -1) | grep -E "$rex" | cut -c1-60,73-120
done; done
