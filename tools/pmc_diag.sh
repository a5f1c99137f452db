#!/bin/bash
# usage (on the GPU box): tools/pmc_diag.sh <tag> [config] -> gpurun_out/<tag>_pmc_diag_<config>.txt
# What a wave of k_match_pairs waits for: instruction-cache and scalar-cache misses, LDS conflicts, memory latency levels
tag=${1:-r03}; cfg=${2:-C1}
R=${GRAFT_REPO_ROOT:-$(pwd)}
out=$R/gpurun_out/pmcdiag_${tag}_$cfg
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
n=0
for set in "SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU SQ_INST_CYCLES_SMEM SQ_IFETCH" \
           "SQC_ICACHE_REQ SQC_ICACHE_MISSES SQC_DCACHE_REQ SQC_DCACHE_MISSES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_ATOMIC_RETURN" \
           "SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_SMEM SQ_INST_LEVEL_LDS SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_IFETCH_LEVEL SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU" \
           "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE"; do
  timeout 200 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $out/$n -o p -- python $R/bench.py --config $cfg --steps 2 --warmup 1 --no-cpu-baseline --no-cold > $out/pass$n.log 2>&1
  n=$((n+1))
done
cd $R
python tools/pmc_summary.py $out 3 2>/dev/null | grep -E "k_match_pairs|k_lists<1|k_inv_records|k_edges" > $R/gpurun_out/${tag}_pmc_diag_$cfg.txt
cat $R/gpurun_out/${tag}_pmc_diag_$cfg.txt
