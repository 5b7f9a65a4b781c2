#!/bin/bash
# Developer aid (CPU only; GPU AddressSanitizer is not available on the pool): host library and oracle built with ASan + UBSan into a scratch
# directory, then tools/sanitize_host.py loads every shipped scene and 40 generated ones, renders them with the oracle and exercises the writers.
set -e
cd "$(dirname "$0")/.."
out=${TMPDIR:-/tmp}/hexray_sanitized; mkdir -p $out
g++ -std=c++17 -O1 -g -fsanitize=address,undefined -fno-omit-frame-pointer -march=haswell -ffast-math -fPIC -Iinclude -Ihexray_amd/host -shared -o $out/libhexray_host.so \
    hexray_amd/host/scene_file.cpp hexray_amd/host/image_io.cpp hexray_amd/host/exr_reader.cpp hexray_amd/host/exr_writer.cpp hexray_amd/host/mesh_prep.cpp \
    hexray_amd/host/heightfield_prep.cpp hexray_amd/host/flatten.cpp hexray_amd/host/host_api.cpp
gcc -std=c11 -O1 -g -fsanitize=address,undefined -fno-omit-frame-pointer -march=haswell -fopenmp -fPIC -Iinclude -shared -o $out/liboracle.so oracle/hxo_oracle.c -lm
LD_PRELOAD="$(gcc -print-file-name=libasan.so) $(gcc -print-file-name=libubsan.so)" ASAN_OPTIONS=detect_leaks=0:halt_on_error=1 UBSAN_OPTIONS=print_stacktrace=1 \
    OMP_NUM_THREADS=4 HXR_SANITIZED_DIR=$out python3 tools/sanitize_host.py
