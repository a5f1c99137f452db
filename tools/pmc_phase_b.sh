#!/bin/bash
# usage (on the GPU box): tools/pmc_phase_b.sh <tag, e.g. r06> [config ...] -> gpurun_out/<tag>_pmc_lists_<config>.json
# The roofline of PHASE B (round 6): per kernel of `bench.py --config C` the HBM-side traffic (FETCH_SIZE and WRITE_SIZE in
# passes of their own, MI355X_MICROARCH.md "rocprofv3 PMC slots"), the instruction and wait counters, the L2 hit rate, and --
# from a separate rocprofv3 --kernel-trace run without counters -- the undisturbed kernel durations.
tag=${1:-r06}; shift; cfgs=${@:-C1 C2 C4}
R=${GRAFT_REPO_ROOT:-$(pwd)}
for cfg in $cfgs; do
  out=$R/gpurun_out/pmcb_${tag}_$cfg
  rm -rf $out; mkdir -p $out
  cd /tmp && export TMPDIR=/tmp
  n=0
  for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE" \
             "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_BUSY_CYCLES SQ_INST_LEVEL_VMEM SQ_THREAD_CYCLES_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SMEM" \
             "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
    timeout 240 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $out/$n -o p -- python $R/bench.py --config $cfg --steps 2 --warmup 1 --no-cpu-baseline --no-cold > $out/pass$n.log 2>&1
    n=$((n+1))
  done
  timeout 240 rocprofv3 --kernel-trace --stats -d $out/trace -o t -- python $R/bench.py --config $cfg --steps 5 --warmup 2 --no-cpu-baseline --no-cold > $out/trace.log 2>&1
  cd $R
  python tools/pmc_phase_b_json.py $out $cfg $tag
done
