#!/bin/bash
# usage (on the GPU box): tools/valu_calib.sh <round tag>  ->  gpurun_out/<tag>_valu_calibration.json
# Runs tools/valu_calib.hip (streams of one known VALU instruction each) plainly -- cycles per instruction from s_memtime
# and from wall time -- and under three rocprofv3 PMC passes, so that what SQ_INSTS_VALU, SQ_ACTIVE_INST_VALU,
# SQ_BUSY_CYCLES and the per-class instruction counters report is known for streams of known content.
tag=${1:-r03}
R=${GRAFT_REPO_ROOT:-$(pwd)}
out=$R/gpurun_out/valu_$tag
mkdir -p $out
bin=$R/tools/bin/valu_calib
[ -x $bin ] || { mkdir -p $R/tools/bin; /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -o $bin $R/tools/valu_calib.hip || exit 1; }
timeout 300 $bin > $out/plain.json 2> $out/plain.err
cd /tmp && export TMPDIR=/tmp
n=0
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY" \
           "SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64" \
           "SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU GRBM_GUI_ACTIVE"; do
  timeout 200 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $out/p$n -o p -- $bin --pmc > $out/pass$n.log 2>&1
  n=$((n+1))
done
cd $R
python tools/valu_calib_json.py $out $tag
