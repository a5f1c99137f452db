#!/bin/bash
# usage (on the GPU box): tools/pmc_keepall.sh <tag> [config] -> gpurun_out/<tag>_pmc_keepall.txt
# rocprofv3 PMC passes (kernel trace only) of tools/keepall_prof.py: the keep-all mode's kernels, averages per launch
tag=${1:-r06}; cfg=${2:-C1}
R=${GRAFT_REPO_ROOT:-$(pwd)}
out=$R/gpurun_out/pmcka_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
n=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" \
           "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_SALU SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_THREAD_CYCLES_VALU" \
           "FETCH_SIZE WRITE_SIZE" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum" "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum"; do
  timeout 200 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $out/$n -o p -- python $R/tools/keepall_prof.py $cfg > $out/pass$n.log 2>&1
  n=$((n+1))
done
cd $R
python - "$out" <<'PY' | tee $R/gpurun_out/${tag}_pmc_keepall.txt
import csv, glob, collections, sys
acc = collections.defaultdict(lambda: collections.defaultdict(float)); calls = collections.defaultdict(set)
for f in sorted(glob.glob(sys.argv[1] + "/*/**/p_counter_collection.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][-40:]
        if not any(x in k for x in ("k_keep", "k_match_pairs", "k_scan")): continue
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); calls[(k, r["Counter_Name"])].add(r["Dispatch_Id"])
for k, v in acc.items():
    print(k)
    for a, b in sorted(v.items()): print(f"    {a:32s} {b / max(len(calls[(k, a)]), 1):16.0f}   ({len(calls[(k, a)])} launches)")
PY
rm -rf $out/[0-9]*
