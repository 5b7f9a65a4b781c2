#!/bin/bash
# Second PMC pass set: instruction-cache, scalar-cache and memory-latency counters (counters only, no tracing domains).
# usage (GPU box, repo root): tools/pmc2.sh <workload> <outdir>
wl=${1:-meshes16}; out=${2:-gpurun_out/pmc2}
root=$(pwd); mkdir -p $out
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAVE_CYCLES SQ_BUSY_CYCLES" \
           "SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQC_DCACHE_MISSES_DUPLICATE SQ_INSTS_SMEM SQ_INST_LEVEL_SMEM SQ_INST_CYCLES_SMEM SQ_INST_CYCLES_SALU" \
           "SQ_INSTS_VMEM SQ_INST_LEVEL_VMEM SQ_INSTS_LDS SQ_INST_LEVEL_LDS SQ_INSTS_FLAT SQ_INSTS_VSKIPPED SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_FLAT" \
           "SQC_TC_INST_REQ SQC_TC_DATA_READ_REQ SQC_TC_REQ SQC_TC_STALL SQC_ICACHE_BUSY_CYCLES SQC_DCACHE_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_LEVEL_WAVES"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $set --output-format csv -d $root/$out/pass$i -- python3 $root/tools/run_frame.py $wl 2 > $root/$out/pass$i.log 2>&1 || echo "pass $i failed"
done
cd $root
python3 tools/pmc_summary.py $out
