#!/bin/bash
# A/B of (library, environment) pairs with tools/r4_ab.py, twice.  usage: tools/r4_ab2.sh <tag> "<lib-suffix or ->|ENV=1 ENV2=2" ...
out=gpurun_out/${1:-r4ab}; mkdir -p $out; shift
for rep in 1 2; do for spec in "$@"; do
	lib=${spec%%|*}; envs=${spec#*|}
	[ "$lib" = "-" ] && path=hexray_amd/lib/libhexray_hip.so || path=hexray_amd/lib/libhexray_hip_$lib.so
	[ -f "$path" ] || { echo "$spec: no $path" >> $out/ab.txt; continue; }
	echo -n "[$envs] " >> $out/ab.txt
	env $envs HXR_HIP_LIB=$(pwd)/$path timeout -k 10 200 python3 tools/r4_ab.py $HXR_AB_WORKLOAD 2>/dev/null | grep steady >> $out/ab.txt || echo "$spec failed" >> $out/ab.txt
done; done
cat $out/ab.txt
