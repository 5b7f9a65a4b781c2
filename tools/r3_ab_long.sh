#!/bin/bash
# like tools/r3_ab.sh, with 14 frames per run (a schedule that takes a few frames to settle shows here)
tag=${1:-ab}; shift
out=gpurun_out/$tag; mkdir -p $out
for w in ${@:-meshes16}; do
  for lib in hexray_amd/lib/libhexray_hip.so hexray_amd/lib/libhexray_hip_*.so; do
    [ -f "$lib" ] || continue
    echo -n "$(basename $lib): " >> $out/ab.txt; HXR_HIP_LIB=$(pwd)/$lib timeout -k 10 200 python3 tools/run_frame.py $w 14 2>&1 | grep "kernel ms" >> $out/ab.txt || echo failed >> $out/ab.txt
  done
done
cat $out/ab.txt
