#!/bin/bash
# A/B of kernel builds inside ONE gpurun call (different boxes differ by ~10%): each lib is timed twice, interleaved
for rep in 1 2; do for lib in hexray_amd/lib/libhexray_hip.so hexray_amd/lib/libhexray_hip_*.so; do
  [ -f "$lib" ] || continue
  echo -n "$(basename $lib): "; HXR_HIP_LIB=$lib timeout -k 10 200 python3 tools/run_frame.py ${1:-meshes16} 5 2>&1 | grep "kernel ms" || echo failed
done; done
