#!/bin/bash
# A/B of every libhexray_hip*.so in hexray_amd/lib with tools/r4_ab.py, twice (box drift shows as the difference between the two rounds)
out=gpurun_out/${1:-r4ab}; mkdir -p $out; shift
for rep in 1 2; do for lib in hexray_amd/lib/libhexray_hip.so hexray_amd/lib/libhexray_hip_*.so; do
	[ -f "$lib" ] || continue
	HXR_HIP_LIB=$(pwd)/$lib timeout -k 10 200 python3 tools/r4_ab.py "$@" 2>/dev/null | grep steady >> $out/ab.txt || echo "$lib failed" >> $out/ab.txt
done; done
cat $out/ab.txt
