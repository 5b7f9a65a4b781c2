#!/bin/bash
for lib in hexray_amd/lib/libhexray_hip.so hexray_amd/lib/libhexray_hip_rev.so hexray_amd/lib/libhexray_hip_prev.so; do for accel in ref sah; do
  echo "== $lib HXR_ACCEL=$accel"
  HXR_HIP_LIB=$PWD/$lib HXR_ACCEL=$accel timeout -k 10 200 python3 tools/run_frame.py meshes16 4
  HXR_HIP_LIB=$PWD/$lib HXR_ACCEL=$accel timeout -k 10 200 python3 tools/shard_time.py meshes16 | grep "world 8\|world 2"
done; done
