#!/bin/bash
# PMC counters of a single-tile render (HXR_SHARD=rank,count exported by the caller)
out=${1:-gpurun_out/pmc3}; root=$(pwd); mkdir -p $out
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_LEVEL_VMEM SQ_INSTS_SMEM SQ_INST_LEVEL_SMEM SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_WAVES"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $set --output-format csv -d $root/$out/pass$i -- python3 $root/tools/run_frame.py meshes16 2 > $root/$out/pass$i.log 2>&1 || echo "pass $i failed"
done
cd $root
python3 tools/pmc_summary.py $out
