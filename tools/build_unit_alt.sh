#!/bin/bash
# Fast A/B builds: recompile ONE kernel unit with other flags / defines and link it with the other objects of the last ./build.sh run.
# usage: UNIT_FLAGS="-O1 ..." tools/build_unit_alt.sh <tag> <unit: mesh | mesh_mc | mesh_lat | mesh2 | mesh_gi | heightfield | nested> [-DNAME=VALUE ...]   (then tools/r3_ab.sh on the GPU box)
set -e
cd "$(dirname "$0")/.."
tag=$1; unit=$2; shift 2
mkdir -p build/alt_$tag
hipcc --offload-arch=gfx950 -std=c++17 -fPIC -fno-slp-vectorize -Iinclude -Ihexray_amd/csrc ${UNIT_FLAGS:--O2 -mllvm -amdgpu-sched-strategy=max-ilp} "$@" -c -o build/alt_$tag/unit.o hexray_amd/csrc/hxr_kernels_$unit.hip
objs="build/hxr_api.o build/hxr_kd_build.o"
for u in mesh mesh_mc mesh_lat mesh2 mesh_gi heightfield nested; do [ $u = $unit ] && objs="$objs build/alt_$tag/unit.o" || objs="$objs build/hxr_kernels_$u.o"; done
hipcc --offload-arch=gfx950 -shared -o hexray_amd/lib/libhexray_hip_$tag.so $objs
python3 tools/kernel_resources.py build/alt_$tag/unit.o | grep "<false, .*, 1>" | cut -c1-140
