#!/bin/bash
# A/B of (library, environment) pairs on bench workloads through tools/run_frame.py (kernel ms of 8 frames).  usage: tools/r4_ab_wl.sh <tag> "<workloads>" "<lib-suffix or ->|ENV=1" ...
out=gpurun_out/${1:-r4wl}; mkdir -p $out; wls=$2; shift 2
for w in $wls; do for rep in 1 2; do for spec in "$@"; do
	lib=${spec%%|*}; envs=${spec#*|}
	[ "$lib" = "-" ] && path=hexray_amd/lib/libhexray_hip.so || path=hexray_amd/lib/libhexray_hip_$lib.so
	echo -n "[$lib $envs] " >> $out/wl.txt
	env $envs HXR_HIP_LIB=$(pwd)/$path timeout -k 10 200 python3 tools/run_frame.py $w 8 2>&1 | grep "kernel ms" >> $out/wl.txt || echo "$w $spec failed" >> $out/wl.txt
done; done; done
cat $out/wl.txt
