#!/bin/bash
# Fast A/B builds of the heightfield translation unit only (see tools/build_mesh_alt.sh).  usage: tools/build_hf_alt.sh <tag> [-D...]
set -e
cd "$(dirname "$0")/.."
tag=$1; shift
mkdir -p build/alt_$tag
hipcc --offload-arch=gfx950 -std=c++17 -fPIC -fno-slp-vectorize -Iinclude -Ihexray_amd/csrc ${HF_FLAGS:--O1} "$@" -c -o build/alt_$tag/hf.o hexray_amd/csrc/hxr_kernels_heightfield.hip
hipcc --offload-arch=gfx950 -shared -o hexray_amd/lib/libhexray_hip_$tag.so build/hxr_api.o build/hxr_kd_build.o build/hxr_kernels_mesh.o build/hxr_kernels_mesh_mc.o build/hxr_kernels_mesh_lat.o build/hxr_kernels_mesh2.o build/hxr_kernels_mesh_gi.o build/alt_$tag/hf.o build/hxr_kernels_nested.o
python3 tools/kernel_resources.py build/alt_$tag/hf.o | grep "<false, .*, 1>" | cut -c1-140
