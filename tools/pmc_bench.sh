#!/bin/bash
# usage (on the GPU box): tools/pmc_bench.sh <round tag, e.g. r02> [config]
# Measures what bench.py's roofline block quotes for the dominant kernel (k_match_pairs) of `bench.py --config C`:
#   * rocprofv3 PMC passes (one per counter set, --kernel-trace only; FETCH_SIZE and WRITE_SIZE in passes of their own,
#     MI355X_MICROARCH.md "rocprofv3 PMC slots")
#   * the candidate counters of the diagnostic build (-DL3D_STATS): pair tests that reach the pre-filter / the exact test
# and writes profiles/<tag>_pmc_match.json keyed by the library's build id (l3d_build_info): bench.py refuses the file
# when the id differs from the build it is timing.
tag=${1:-r03}; cfg=${2:-C1}
R=${GRAFT_REPO_ROOT:-$(pwd)}
out=$R/gpurun_out/pmc_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
n=0
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" \
           "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" \
           "SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64" \
           "SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  timeout 150 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $out/$n -o p -- python $R/bench.py --config $cfg --steps 2 --warmup 1 --no-cpu-baseline --no-cold > $out/pass$n.log 2>&1
  n=$((n+1))
done
cd $R
# the diagnostic build travels with the snapshot when it was made beforehand (tools/build_stats_lib.sh); its build id is
# compared with the product library's in pmc_to_json.py
[ -f $R/gpurun_scratch/libl3dpp_hip_stats.so ] || bash tools/build_stats_lib.sh > $out/stats_build.log 2>&1
L3D_LIB=$R/gpurun_scratch/libl3dpp_hip_stats.so python tools/phase_a_stats.py $cfg > $out/stats.json 2> $out/stats.err
python tools/pmc_to_json.py $out $cfg $tag
