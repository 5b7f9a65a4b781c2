#!/bin/bash
# A/B of kernel builds on the 8-way shard bound (tools/shard_time.py) and the whole frame, inside ONE gpurun call
for lib in hexray_amd/lib/libhexray_hip.so hexray_amd/lib/libhexray_hip_*.so; do
  [ -f "$lib" ] || continue
  echo "== $(basename $lib)"
  HXR_HIP_LIB=$(pwd)/$lib HXR_TILE=8 HXR_ALL_RANKS=1 timeout -k 10 300 python3 tools/shard_time.py ${1:-meshes16} 2>&1 | grep "world [18] " | awk '{print $1,$2,$3,$4,$9,$10,$11,$12}'
done
