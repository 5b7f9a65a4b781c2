#!/bin/bash
# Round profile: kernel trace + stats of the default bench, PMC passes for HBM traffic; outputs under gpurun_out/<tag>
tag=${1:-r01_final}
root=$(pwd); out=$root/gpurun_out/$tag; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 $root/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-sah-side > $out/bench.json 2> $out/bench.err
cd $root
tools/pmc.sh meshes16 gpurun_out/$tag/pmc > $out/pmc.txt 2>&1
find $out/trace -name "*kernel_stats.csv" -exec cp {} $out/kernel_stats.csv \;
cat $out/bench.json; head -4 $out/kernel_stats.csv; cat $out/pmc.txt | tail -32
