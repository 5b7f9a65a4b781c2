#!/bin/bash
# Build the render kernel with in-kernel phase clocks and print where the waves' time goes.
# usage (repo root): tools/phase_profile.sh build            (anywhere: cross-compiles both phase sets)
#                    tools/phase_profile.sh [workload ...]   (GPU box)
set -e
lib1=hexray_amd/lib/libhexray_hip_timers1.so; lib2=hexray_amd/lib/libhexray_hip_timers2.so; lib3=hexray_amd/lib/libhexray_hip_timers3.so
if [ "$1" = build ] || [ ! -f $lib1 ] || [ ! -f $lib2 ] || [ ! -f $lib3 ]; then
	for set in 1 2 3; do
		hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DHXR_PHASE_TIMERS -DHXR_PHASE_SET=$set -Iinclude -Ihexray_amd/csrc \
			-o hexray_amd/lib/libhexray_hip_timers$set.so hexray_amd/csrc/hxr_api.hip hexray_amd/csrc/hxr_kernels_mesh.hip hexray_amd/csrc/hxr_kernels_heightfield.hip
	done
	[ "$1" = build ] && exit 0
fi
mkdir -p gpurun_out
{
HXR_HIP_LIB=$(pwd)/$lib1 timeout -k 10 300 python3 tools/phase_profile.py "$@"
HXR_PHASE_NAMES="dispense,gen,reduce+store,query(all),answer(lights/env/surface),shaders+query prep" HXR_HIP_LIB=$(pwd)/$lib2 timeout -k 10 300 python3 tools/phase_profile.py "$@"
HXR_PHASE_NAMES="dispense,pixel map,jitter+lens,camera+trace setup,reduce+store,trace_rays" HXR_HIP_LIB=$(pwd)/$lib3 timeout -k 10 300 python3 tools/phase_profile.py "$@"
} | tee gpurun_out/phase_profile.txt
