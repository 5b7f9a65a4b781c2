#!/bin/bash
# parity of the element classes outside the five configs
run() { timeout -k 5 200 python tools/gpu_check.py "$@" 2>&1 | grep -v "loadBMP\|amdgpu.ids\|Cubemap" | head -4; }
run data/kdtree_test.hexray 320 240 0 0
run data/hw10/bokeh.hexray 200 150 0 0 "Camera {\nnumSamples 4\n}\n"
run data/meshes.hexray 200 150 1 0 "Reflection reflection {\nglossiness 0.6\nnumSamples 5\n}\n"
run data/meshes.hexray 200 150 0 0 "Camera {\ndof on\nautoFocus on\nnumSamples 4\n}\n"
run data/boxed.hexray 160 120 0 0 "GlobalSettings {\nnumPaths 4\n}\nCamera {\nnumSamples 4\n}\n"
run data/cornell_box.hexray 160 120 0 0 "GlobalSettings {\nnumPaths 4\n}\nCamera {\nnumSamples 4\n}\n"
run data/simple.hexray 640 360 1 0
run data/heightfield.hexray 480 270 2 0
