#!/bin/bash
# Build the render kernels with in-kernel phase clocks (same per-unit flags as build.sh, through tools/build_alt.sh) and print where
# the waves' time goes.
# usage (repo root): tools/phase_profile.sh build            (anywhere: cross-compiles the three phase sets)
#                    tools/phase_profile.sh [workload ...]   (GPU box)
set -e
lib1=hexray_amd/lib/libhexray_hip_timers1.so; lib2=hexray_amd/lib/libhexray_hip_timers2.so; lib3=hexray_amd/lib/libhexray_hip_timers3.so
if [ "$1" = build ] || [ ! -f $lib1 ] || [ ! -f $lib2 ] || [ ! -f $lib3 ]; then
	for set in 1 2 3; do tools/build_alt.sh timers$set -DHXR_PHASE_TIMERS -DHXR_PHASE_SET=$set; done
	[ "$1" = build ] && exit 0
fi
mkdir -p gpurun_out
{
HXR_HIP_LIB=$(pwd)/$lib1 timeout -k 10 300 python3 tools/phase_profile.py "$@"
HXR_PHASE_NAMES="dispense,gen,reduce+store,query(all),answer(lights/env/surface),shaders+query prep" HXR_HIP_LIB=$(pwd)/$lib2 timeout -k 10 300 python3 tools/phase_profile.py "$@"
HXR_PHASE_NAMES="dispense,pixel map,jitter+lens,camera+trace setup,reduce+store,trace_rays" HXR_HIP_LIB=$(pwd)/$lib3 timeout -k 10 300 python3 tools/phase_profile.py "$@"
} | tee gpurun_out/phase_profile.txt
