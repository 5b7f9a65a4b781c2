#!/bin/bash
# Fast A/B builds: recompile ONLY the unit of the headline kernel (hxr_kernels_mesh_mc.hip; MESH_UNIT=mesh: hxr_kernels_mesh.hip) with extra defines and link it with the
# other objects of the last ./build.sh run.  usage: tools/build_mesh_alt.sh <tag> [-DNAME=VALUE ...]     (then tools/r3_ab.sh on the GPU box)
set -e
cd "$(dirname "$0")/.."
tag=$1; shift
mkdir -p build/alt_$tag
unit=${MESH_UNIT:-mesh_mc}
if [ $unit = mesh_mc ]; then flags=${MESH_FLAGS:--O1 -mllvm -amdgpu-sched-strategy=max-memory-clause}; other=build/hxr_kernels_mesh.o; else flags=${MESH_FLAGS:--O2 -mllvm -amdgpu-sched-strategy=max-ilp}; other=build/hxr_kernels_mesh_mc.o build/hxr_kernels_mesh_lat.o; fi
hipcc --offload-arch=gfx950 -std=c++17 -fPIC -fno-slp-vectorize -Iinclude -Ihexray_amd/csrc $flags "$@" -c -o build/alt_$tag/mesh.o hexray_amd/csrc/hxr_kernels_$unit.hip
hipcc --offload-arch=gfx950 -shared -o hexray_amd/lib/libhexray_hip_$tag.so build/hxr_api.o build/hxr_kd_build.o build/alt_$tag/mesh.o $other build/hxr_kernels_mesh2.o build/hxr_kernels_mesh_gi.o build/hxr_kernels_heightfield.o build/hxr_kernels_nested.o
python3 tools/kernel_resources.py build/alt_$tag/mesh.o | grep "<false, 1u, [0-9], 1>" | cut -c1-140
