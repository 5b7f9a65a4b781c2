"""SURVEY.md 8(f) rank 1 on the CPU: the SAH kd-tree the module builds (hxr_sah_build, host-only) is handed to the ORACLE in
place of the reference's tree.  The oracle's walk is the reference's recursion, so an identical frame shows that the tree keeps
the property the walk relies on: every triangle is listed in every leaf whose box it reaches."""
import ctypes

import numpy as np
import pytest

import hexray_amd as hx
import desc_types as dt
from conftest import scene_path
from oracle_binding import oracle_lib


def sah_desc(hs):
    """A copy of the scene descriptor whose meshes carry the SAH trees; returns (desc copy, keep-alive list)."""
    lib = ctypes.CDLL(hx.HIP_LIB_PATH)
    lib.hxr_sah_build.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_int64,
                                  ctypes.POINTER(ctypes.c_int64), ctypes.POINTER(ctypes.c_int64), ctypes.POINTER(ctypes.c_int)]
    src = dt.desc_of(hs)
    trees, stats = [], []
    for mi in range(src.num_meshes):
        n, l, dpt = ctypes.c_int64(), ctypes.c_int64(), ctypes.c_int()
        assert lib.hxr_sah_build(hs.desc, mi, None, 0, None, 0, ctypes.byref(n), ctypes.byref(l), ctypes.byref(dpt)) == 0
        nodes = (dt.KdNode * n.value)()
        leaves = (ctypes.c_int32 * max(1, l.value))()
        assert lib.hxr_sah_build(hs.desc, mi, nodes, n.value, leaves, l.value, ctypes.byref(n), ctypes.byref(l), ctypes.byref(dpt)) == 0
        trees.append((nodes, leaves, n.value, l.value))
        stats.append((n.value, l.value, dpt.value))
    all_nodes = (dt.KdNode * sum(t[2] for t in trees))()
    all_leaves = (ctypes.c_int32 * max(1, sum(t[3] for t in trees)))()
    meshes = (dt.Mesh * src.num_meshes)()
    no, lo = 0, 0
    for mi, (nodes, leaves, n, l) in enumerate(trees):
        ctypes.memmove(ctypes.byref(meshes[mi]), ctypes.byref(src.meshes[mi]), ctypes.sizeof(dt.Mesh))
        if src.meshes[mi].has_kd:
            ctypes.memmove(ctypes.byref(all_nodes, no * ctypes.sizeof(dt.KdNode)), nodes, n * ctypes.sizeof(dt.KdNode))
            ctypes.memmove(ctypes.byref(all_leaves, lo * 4), leaves, l * 4)
            meshes[mi].first_kd_node, meshes[mi].num_kd_nodes = no, n
            meshes[mi].first_leaf_index, meshes[mi].num_leaf_indices = lo, l
            no, lo = no + n, lo + l
    d = dt.SceneDesc()
    ctypes.memmove(ctypes.byref(d), ctypes.byref(src), ctypes.sizeof(dt.SceneDesc))
    d.meshes = ctypes.cast(meshes, ctypes.POINTER(dt.Mesh))
    d.kd_nodes, d.num_kd_nodes = ctypes.cast(all_nodes, ctypes.POINTER(dt.KdNode)), no
    d.leaf_indices, d.num_leaf_indices = ctypes.cast(all_leaves, ctypes.POINTER(ctypes.c_int32)), lo
    return d, (all_nodes, all_leaves, meshes), stats


def render_with(desc_addr, hs, params):
    s = hs.settings
    w, h = params.width or s.frame_width, params.height or s.frame_height
    out = np.zeros((h, w, 3), dtype=np.float32)
    cnt = hx.Counters()
    assert oracle_lib().hxo_render(desc_addr, ctypes.byref(params), out.ctypes.data, w * 12, 0, ctypes.byref(cnt)) == 0
    return out, cnt.as_dict()


@pytest.mark.parametrize("scene,ov,fp", [
    ("meshes.hexray", hx.overlay_text(200, 112, want_aa=False), dict(mode=hx.MODE_SINGLE)),
    ("meshes.hexray", hx.overlay_text(120, 68, dof=True, num_samples=4), dict()),
    ("simple.hexray", hx.overlay_text(200, 150, want_aa=False), dict(mode=hx.MODE_SINGLE)),
    ("cornell_box.hexray", hx.overlay_text(64, 48, num_paths=2), dict()),
])
def test_oracle_renders_the_same_frame_through_the_sah_trees(scene, ov, fp):
    hs = hx.HostScene(scene_path(scene), ov)
    ref, rc = render_with(hs.desc, hs, hx.frame_params(**fp))
    d, keep, stats = sah_desc(hs)
    sah, sc = render_with(ctypes.addressof(d), hs, hx.frame_params(**fp))
    assert np.array_equal(ref, sah), float(np.abs(ref - sah).max())
    for k in ("camera_samples", "raycasts", "shadow_queries", "node_tests", "env_bilinear", "tex_bilinear"):
        assert rc[k] == sc[k], (k, rc[k], sc[k])
    if rc["tri_tests"] > 100000:
        assert sc["tri_tests"] <= rc["tri_tests"]   # and it is never the more expensive tree to walk
    for n, l, depth in stats:
        assert n % 2 == 1 and depth <= 40


def test_sah_builder_rejects_bad_arguments():
    hs = hx.HostScene(scene_path("meshes.hexray"), hx.overlay_text(32, 24))
    lib = ctypes.CDLL(hx.HIP_LIB_PATH)
    lib.hxr_sah_build.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_int64,
                                  ctypes.POINTER(ctypes.c_int64), ctypes.POINTER(ctypes.c_int64), ctypes.POINTER(ctypes.c_int)]
    assert lib.hxr_sah_build(hs.desc, 99, None, 0, None, 0, None, None, None) < 0
    n, l = ctypes.c_int64(), ctypes.c_int64()
    assert lib.hxr_sah_build(hs.desc, 1, None, 0, None, 0, ctypes.byref(n), ctypes.byref(l), None) == 0 and n.value > 1
    small = (dt.KdNode * 1)()
    leaves = (ctypes.c_int32 * 1)()
    assert lib.hxr_sah_build(hs.desc, 1, small, 1, leaves, 1, None, None, None) < 0
