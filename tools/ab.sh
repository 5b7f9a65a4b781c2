#!/bin/bash
# A/B kernel variants: every hexray_amd/lib/libhexray_hip_*.so is timed on the same workload in turn.
wl=${1:-meshes16}
for lib in hexray_amd/lib/libhexray_hip.so hexray_amd/lib/libhexray_hip_*.so; do
  [ -f "$lib" ] || continue
  echo -n "$(basename $lib): "
  HXR_HIP_LIB=$lib timeout -k 10 200 python3 tools/run_frame.py $wl 4 2>&1 | grep "kernel ms" || echo failed
done
