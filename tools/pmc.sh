#!/bin/bash
# Collect PMC counters for one workload in separate rocprofv3 passes (counters only, no tracing domains).
# usage (on the GPU box, from the repo root): tools/pmc.sh <workload> <outdir>
wl=${1:-meshes16}; out=${2:-gpurun_out/pmc}
root=$(pwd); mkdir -p $out
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU" \
           "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64" \
           "SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT SQ_WAVES SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_INSTS_VMEM_WR" \
           "FETCH_SIZE TCC_REQ_sum" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $set --output-format csv -d $root/$out/pass$i -- python3 $root/tools/run_frame.py $wl 2 > $root/$out/pass$i.log 2>&1 || echo "pass $i failed"
done
cd $root
python3 tools/pmc_summary.py $out
