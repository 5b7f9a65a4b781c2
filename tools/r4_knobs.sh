#!/bin/bash
# a workload under schedule knobs: median kernel ms of 12 steady frames, each setting twice.  usage: tools/r4_knobs.sh <outdir> "<workloads>" "ENV=V" ...
out=gpurun_out/$1; wls=$2; shift 2; mkdir -p $out
for w in $wls; do for rep in 1 2; do for spec in "$@"; do
	echo -n "$w [$spec] " >> $out/knobs.txt
	env $spec timeout -k 10 300 python3 tools/run_frame.py $w 16 2>&1 | grep "kernel ms" | python3 -c "import sys,statistics; v=[float(x) for x in sys.stdin.read().split(':')[1].split()][4:]; print('median %.3f  min %.3f  max %.3f' % (statistics.median(v), min(v), max(v)))" >> $out/knobs.txt
done; done; done
cat $out/knobs.txt
