#!/bin/bash
# Build an alternative libhexray_hip_<tag>.so with extra compiler defines, for an A/B inside one gpurun call (tools/ab2.sh).
# usage: tools/build_alt.sh <tag> [-DNAME=VALUE ...]
set -e
cd "$(dirname "$0")/.."
tag=$1; shift
mkdir -p build/alt_$tag
HIPFLAGS="--offload-arch=gfx950 -std=c++17 -fPIC -fno-slp-vectorize -Iinclude -Ihexray_amd/csrc $*"
MESH_FLAGS=${MESH_FLAGS:--O2 -mllvm -amdgpu-sched-strategy=max-ilp}   # flags of the kd-walking translation units (override for compiler-flag A/Bs)
hipcc $HIPFLAGS $MESH_FLAGS -c -o build/alt_$tag/mesh.o hexray_amd/csrc/hxr_kernels_mesh.hip &
p1=$!
hipcc $HIPFLAGS ${HF_FLAGS:--O1} -c -o build/alt_$tag/hf.o hexray_amd/csrc/hxr_kernels_heightfield.hip &
p2=$!
hipcc $HIPFLAGS ${MESH2_FLAGS:--O2 -mllvm -amdgpu-sched-strategy=max-ilp} -c -o build/alt_$tag/mesh2.o hexray_amd/csrc/hxr_kernels_mesh2.hip &
p3=$!
hipcc $HIPFLAGS ${MESH_GI_FLAGS:--O1 -mllvm -amdgpu-sched-strategy=max-memory-clause} -c -o build/alt_$tag/mesh_gi.o hexray_amd/csrc/hxr_kernels_mesh_gi.hip &
p6=$!
hipcc $HIPFLAGS -O2 -c -o build/alt_$tag/nested.o hexray_amd/csrc/hxr_kernels_nested.hip &
p4=$!
hipcc $HIPFLAGS ${MESH_MC_FLAGS:--O1 -mllvm -amdgpu-sched-strategy=max-ilp -mllvm -amdgpu-use-amdgpu-trackers=1} -c -o build/alt_$tag/mesh_mc.o hexray_amd/csrc/hxr_kernels_mesh_mc.hip &
p5=$!
hipcc $HIPFLAGS ${MESH_LAT_FLAGS:--O1 -mllvm -amdgpu-sched-strategy=max-ilp -mllvm -amdgpu-use-amdgpu-trackers=1} -c -o build/alt_$tag/mesh_lat.o hexray_amd/csrc/hxr_kernels_mesh_lat.hip &
p7=$!
hipcc $HIPFLAGS -O3 -c -o build/alt_$tag/kd.o hexray_amd/csrc/hxr_kd_build.hip
hipcc $HIPFLAGS ${API_FLAGS:--O3 -mllvm -amdgpu-sched-strategy=max-memory-clause} -c -o build/alt_$tag/api.o hexray_amd/csrc/hxr_api.hip
wait $p1; wait $p2; wait $p3; wait $p4; wait $p5; wait $p6; wait $p7
hipcc --offload-arch=gfx950 -shared -o hexray_amd/lib/libhexray_hip_$tag.so build/alt_$tag/api.o build/alt_$tag/kd.o build/alt_$tag/mesh.o build/alt_$tag/mesh_mc.o build/alt_$tag/mesh_lat.o build/alt_$tag/mesh2.o build/alt_$tag/mesh_gi.o build/alt_$tag/hf.o build/alt_$tag/nested.o
echo built hexray_amd/lib/libhexray_hip_$tag.so
