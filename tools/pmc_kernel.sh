#!/bin/bash
# usage (on the GPU box): tools/pmc_kernel.sh <tag> <kernel-name-substring> "<COUNTERS ...>" ["<COUNTERS ...>" ...]
# rocprofv3 PMC passes (one per counter set, --kernel-trace only) of `bench.py --steps 2 --warmup 1`, summed per launch
# of the kernels whose name contains the substring -> gpurun_out/pmc_<tag>/summary.txt
tag=$1; shift; pat=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
n=0
for set in "$@"; do
  timeout 120 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $R/gpurun_out/pmc_$tag/$n -o p -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
  n=$((n+1))
done
cd $R
python - "$R/gpurun_out/pmc_$tag" "$pat" <<'PY' | tee $R/gpurun_out/pmc_$tag/summary.txt
import csv, glob, collections, sys
acc = collections.defaultdict(lambda: collections.defaultdict(float)); calls = collections.defaultdict(set)
for f in sorted(glob.glob(sys.argv[1] + "/*/**/p_counter_collection.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        if sys.argv[2] not in r["Kernel_Name"]: continue
        k = r["Kernel_Name"].split("(")[0][-48:]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); calls[(k, r["Counter_Name"])].add(r["Dispatch_Id"])
for k, v in acc.items():
    print(k, {a: round(b / max(len(calls[(k, a)]), 1)) for a, b in sorted(v.items())}, "launches", {a: len(calls[(k, a)]) for a in v})
PY
