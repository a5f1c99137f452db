#!/bin/bash
# usage (on the GPU box): tools/prof_bench.sh <tag> [bench args] -> gpurun_out/prof_<tag>/ + summary on stdout
tag=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_$tag -o $tag -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline "$@" > $R/gpurun_out/prof_$tag.log 2>&1
cd $R
python tools/prof_summary.py $(find gpurun_out/prof_$tag -name "*.db" | head -1)
