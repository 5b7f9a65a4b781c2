#!/bin/bash
# Build libhexray_hip_<tag>.so that differs from the shipped library in the headline unit (hxr_kernels_mesh_mc.hip) only: extra defines for that unit,
# every other object taken from build/ (run build.sh first).  usage: tools/build_alt_mc.sh <tag> [-DNAME=VALUE ...]
set -e
cd "$(dirname "$0")/.."
tag=$1; shift
mkdir -p build/alt_$tag
hipcc --offload-arch=gfx950 -std=c++17 -fPIC -fno-slp-vectorize -Iinclude -Ihexray_amd/csrc "$@" -O1 -mllvm -amdgpu-sched-strategy=max-ilp -mllvm -amdgpu-use-amdgpu-trackers=1 -c -o build/alt_$tag/mesh_mc.o hexray_amd/csrc/hxr_kernels_mesh_mc.hip
hipcc --offload-arch=gfx950 -shared -o hexray_amd/lib/libhexray_hip_$tag.so build/hxr_api.o build/hxr_kd_build.o build/hxr_kernels_mesh.o build/alt_$tag/mesh_mc.o build/hxr_kernels_mesh_lat.o build/hxr_kernels_mesh2.o build/hxr_kernels_mesh_gi.o build/hxr_kernels_heightfield.o build/hxr_kernels_nested.o
python3 tools/kernel_resources.py build/alt_$tag/mesh_mc.o | tail -2
echo built hexray_amd/lib/libhexray_hip_$tag.so
