#!/bin/bash
# round 3: A/B of every libhexray_hip*.so in hexray_amd/lib on the workloads given (default: the mesh-walking ones), then the GPU test suite
# usage: tools/r3_ab.sh <outdir-tag> [workload ...]       (HXR_AB_NO_TESTS=1 skips the suite)
tag=${1:-ab}; shift
out=gpurun_out/$tag; mkdir -p $out
wls=${@:-meshes16 cornell16 kdtree16 simple1 meshes_aa}
for w in $wls; do
  for rep in 1 2; do for lib in hexray_amd/lib/libhexray_hip.so hexray_amd/lib/libhexray_hip_*.so; do
    [ -f "$lib" ] || continue
    echo -n "$(basename $lib): " >> $out/ab.txt; HXR_HIP_LIB=$lib timeout -k 10 200 python3 tools/run_frame.py $w 6 2>&1 | grep "kernel ms" >> $out/ab.txt || echo failed >> $out/ab.txt
  done; done
done
cat $out/ab.txt
if [ -z "$HXR_AB_NO_TESTS" ]; then timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > $out/pytest.txt 2>&1; tail -15 $out/pytest.txt; fi
