#!/bin/bash
# usage (on the GPU box): tools/pmc_bench.sh <tag> "<COUNTERS ...>" ["<COUNTERS ...>" ...] -> gpurun_out/pmc_<tag>/<n>/
tag=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
n=0
for set in "$@"; do
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $R/gpurun_out/pmc_$tag/$n -o p -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
  n=$((n+1))
done
cd $R
python tools/pmc_summary.py gpurun_out/pmc_$tag 3
