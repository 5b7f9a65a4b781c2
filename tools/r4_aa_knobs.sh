#!/bin/bash
# adaptive-AA frame under schedule knobs: median kernel ms of 40 frames, each setting twice.  usage: tools/r4_aa_knobs.sh <outdir> "ENV=V" ...
out=gpurun_out/$1; shift; mkdir -p $out
for rep in 1 2; do for spec in "$@"; do
	echo -n "[$spec] " >> $out/aa.txt
	env $spec timeout -k 10 200 python3 tools/run_frame.py meshes_aa 43 2>&1 | grep "kernel ms" | python3 -c "import sys,statistics; v=[float(x) for x in sys.stdin.read().split(':')[1].split()][3:]; print('median %.3f  min %.3f  max %.3f' % (statistics.median(v), min(v), max(v)))" >> $out/aa.txt
done; done
cat $out/aa.txt
