#!/bin/bash
# A/B of kernel builds with tools/exp_defer.py inside ONE gpurun call
for lib in hexray_amd/lib/libhexray_hip.so hexray_amd/lib/libhexray_hip_*.so; do
  [ -f "$lib" ] || continue
  echo "== $(basename $lib)"; HXR_HIP_LIB=$(pwd)/$lib FRAMES=5 timeout -k 10 200 python3 tools/exp_defer.py ${1:-meshes16} 2>&1 | grep -v amdgpu.ids | head -${2:-1} || echo failed
done
