"""A mesh an order of magnitude beyond the reference's largest (a 120 000-triangle bumpy torus, generated here): the host build, the GPU
build (several rounds per workgroup, buffer growth) and the packet walk at a scale the shipped scenes do not reach."""
import math
import os

import numpy as np
import pytest

from conftest import ROOT
from desc_types import SceneDesc
from oracle_binding import oracle_render

pytestmark = pytest.mark.gpu

SCENE = """
GlobalSettings {{
	frameWidth 160
	frameHeight 120
	ambientLight (0.2, 0.2, 0.2)
	wantAA off
	prepassSamples 0
	maxTraceDepth 4
}}
PointLight l {{
	pos (30, 60, -40)
	power 12000
}}
Camera camera {{
	pos (0, 9, -24)
	aspectRatio 1.33333
	yaw 0
	pitch -20
	roll 0
	fov 60
}}
Plane floor {{
	y -4
}}
CheckerTexture checker {{
	color1 (0.8, 0.8, 0.8)
	color2 (0.2, 0.2, 0.4)
	scaling 3
}}
Lambert floor_shader {{
	color (1, 1, 1)
	texture checker
}}
Node floor_node {{
	geometry floor
	shader floor_shader
}}
Mesh torus {{
	file "{obj}"
}}
Phong torus_shader {{
	color (0.9, 0.5, 0.2)
	exponent 40
}}
Reflection mirror {{
	reflColor (0.9, 0.9, 0.9)
}}
Layered torus_layered {{
	layer torus_shader (1, 1, 1)
	layer mirror (0.2, 0.2, 0.2)
}}
Node torus_node {{
	geometry torus
	shader torus_layered
}}
"""


def write_torus(path, nu=300, nv=200):
    """2 * nu * nv triangles; radii perturbed so that no two vertices share a coordinate (no grazing ties for the builders)."""
    with open(path, "w") as f:
        for i in range(nu):
            for j in range(nv):
                u, v = 2 * math.pi * i / nu, 2 * math.pi * j / nv
                r = 3.0 + 0.25 * math.sin(5 * u + 0.3) * math.cos(7 * v + 0.1)
                x, y, z = (9 + r * math.cos(v)) * math.cos(u), r * math.sin(v), (9 + r * math.cos(v)) * math.sin(u)
                f.write("v %.6f %.6f %.6f\n" % (x + 0.013, y + 0.017, z - 0.011))
        for i in range(nu):
            for j in range(nv):
                f.write("vt %.6f %.6f\n" % (i / nu, j / nv))
        for i in range(nu):
            for j in range(nv):
                a, b = i * nv + j + 1, ((i + 1) % nu) * nv + j + 1
                c, d = ((i + 1) % nu) * nv + (j + 1) % nv + 1, i * nv + (j + 1) % nv + 1
                f.write("f %d/%d %d/%d %d/%d\n" % (a, a, b, b, c, c))
                f.write("f %d/%d %d/%d %d/%d\n" % (a, a, c, c, d, d))


@pytest.fixture(scope="module")
def torus_scene(tmp_path_factory):
    d = tmp_path_factory.mktemp("torus")
    obj = os.path.join(d, "torus.obj")
    write_torus(obj)
    scene = os.path.join(d, "torus.hexray")
    with open(scene, "w") as f:
        f.write(SCENE.format(obj="torus.obj"))   # asset paths are relative to the scene file
    return scene


def test_large_mesh_tree_built_on_the_gpu_equals_the_host_build(gpu, torus_scene):
    hx = gpu
    hs = hx.HostScene(torus_scene)
    desc = SceneDesc.from_address(hs.desc)
    m = desc.meshes[0]
    assert m.num_triangles == 120000 and m.num_kd_nodes > 20000
    nodes, leaf, depth, ms = hx.kd_build_device(hs, 0)
    want = np.zeros(m.num_kd_nodes, dtype=hx.KD_NODE_DTYPE)
    import ctypes
    ctypes.memmove(want.ctypes.data, ctypes.addressof(desc.kd_nodes[m.first_kd_node]), want.nbytes)
    want_leaf = np.ctypeslib.as_array(desc.leaf_indices, shape=(desc.num_leaf_indices,))[m.first_leaf_index:m.first_leaf_index + m.num_leaf_indices]
    assert len(nodes) == len(want) and len(leaf) == len(want_leaf)
    for field in ("axis", "first", "count", "parent"):
        assert np.array_equal(nodes[field], want[field]), field
    assert np.array_equal(nodes["split"].view(np.uint64), want["split"].view(np.uint64))
    assert np.array_equal(leaf, want_leaf)
    print("120000 triangles: %d nodes, %d leaf entries, depth %d, GPU build %.1f ms" % (len(nodes), len(leaf), depth, ms))


def test_large_mesh_frame_against_the_oracle(gpu, torus_scene):
    hx = gpu
    hs = hx.HostScene(torus_scene)
    p = hx.frame_params(collect_counters=True)
    want, want_counters = oracle_render(hs, p)
    for accel in (hx.ACCEL_REFERENCE, hx.ACCEL_REFERENCE_DEVICE_BUILD):
        ds = hx.DeviceScene(hs, accel=accel)
        got = ds.render(p)
        rms = np.sqrt(np.mean((got.astype(np.float64) - want) ** 2, axis=(0, 1)))
        assert rms.max() <= 1e-4, (accel, rms)
        c = ds.counters()
        for k in ("camera_samples", "raycasts", "shadow_queries", "node_tests", "bbox_tests", "kd_visits", "leaf_indices", "tri_tests"):
            assert c[k] == want_counters[k], (accel, k, c[k], want_counters[k])
