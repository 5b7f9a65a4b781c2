#!/bin/bash
# Reproducer of the code-generation defect behind round 3's GPU memory fault, WITHOUT a GPU: compile the catch-all unit with the flags that faulted
# and let the lint find the misplaced copy.  Expected output (ROCm 7.2.0, AMD clang 22):
#   unit.o: void hxr_render_kernel<false, 239u, 3, -1> +0x586c: `v_mov_b32_e32 v66, v100` runs ahead of `s_or_b64 exec, exec, s[2:3]`
# (the offset moves with the source; the shipped flags -- max-ilp for this unit -- give 0 findings).  The full-size kernel is the smallest form known: the
# copy is a live-range split made under the register pressure of a 57 000-instruction kernel.
cd "$(dirname "$0")/../.."
mkdir -p build/alt_defect
hipcc --offload-arch=gfx950 -std=c++17 -fPIC -fno-slp-vectorize -Iinclude -Ihexray_amd/csrc -O2 -mllvm -amdgpu-sched-strategy=max-memory-clause \
	-c -o build/alt_defect/unit.o hexray_amd/csrc/hxr_kernels_mesh2.hip
python3 tools/check_exec_prologue.py build/alt_defect/unit.o
