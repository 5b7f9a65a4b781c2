#!/bin/bash
# Round profile: kernel trace + stats of the default bench, then PMC passes (counters only, separate runs); outputs under gpurun_out/<tag>.
# Copy what should be judged into profiles/ afterwards (tools/profile_collect.sh <tag>).
tag=${1:-r03}
root=$(pwd); out=$root/gpurun_out/$tag; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 $root/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-sah-side --no-other-timings --no-live-pmc > $out/bench_under_rocprof.json 2> $out/bench.err
cd $root
find $out/trace -name "*kernel_stats.csv" -exec cp {} $out/kernel_stats.csv \;
rm -rf $out/trace
tools/pmc.sh meshes16 gpurun_out/$tag/pmc > $out/pmc.txt 2>&1
rm -rf $out/pmc/pass*/
cat $out/bench_under_rocprof.json; head -6 $out/kernel_stats.csv; tail -3 $out/pmc.txt
