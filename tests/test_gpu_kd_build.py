"""Mesh::buildKD as GPU kernels (hxr_kd_build_device / HXR_ACCEL_REFERENCE_DEVICE_BUILD).

The device build must produce the tree the host builder produces -- same splits, same leaf lists in the same order -- for every
mesh the reference ships, and a scene created from a descriptor WITHOUT trees must render exactly what the scene with the
host-built trees renders."""
import ctypes
import os

import numpy as np
import pytest

from conftest import ROOT, scene_path
from desc_types import SceneDesc

pytestmark = pytest.mark.gpu
ALL_MESHES = os.path.join(ROOT, "tests", "golden", "all_meshes.hexray")


def host_tree(desc, mi):
    import hexray_amd as hx
    m = desc.meshes[mi]
    nodes = np.zeros(m.num_kd_nodes, dtype=hx.KD_NODE_DTYPE)
    if m.num_kd_nodes:
        ctypes.memmove(nodes.ctypes.data, ctypes.addressof(desc.kd_nodes[m.first_kd_node]), nodes.nbytes)
    leaf = np.array([desc.leaf_indices[m.first_leaf_index + i] for i in range(m.num_leaf_indices)], dtype=np.int32)
    return nodes, leaf


def test_device_kd_build_equals_host_build_for_every_reference_mesh(gpu):
    hx = gpu
    hs = hx.HostScene(ALL_MESHES)
    desc = SceneDesc.from_address(hs.desc)
    assert desc.num_meshes == 9
    for mi in range(desc.num_meshes):
        want_nodes, want_leaf = host_tree(desc, mi)
        nodes, leaf, depth, ms = hx.kd_build_device(hs, mi)
        assert len(nodes) == len(want_nodes) and len(leaf) == len(want_leaf), (mi, len(nodes), len(want_nodes), len(leaf), len(want_leaf))
        for field in ("axis", "first", "count", "parent"):
            assert np.array_equal(nodes[field], want_nodes[field]), (mi, field)
        assert np.array_equal(nodes["split"].view(np.uint64), want_nodes["split"].view(np.uint64)), mi   # bit-exact split positions
        assert np.array_equal(leaf, want_leaf), mi
        assert depth == hs.kd_stats()[mi][2]


def test_scene_without_trees_renders_what_the_host_built_trees_render(gpu):
    hx = gpu
    ov = hx.overlay_text(width=480, height=270, dof=True, num_samples=4)
    with_trees = hx.HostScene(scene_path("meshes.hexray"), ov)
    without = hx.HostScene(scene_path("meshes.hexray"), ov, gpu_kd_build=True)
    d = SceneDesc.from_address(without.desc)
    assert d.num_kd_nodes == 0 and d.num_leaf_indices == 0 and all(d.meshes[i].has_kd for i in range(d.num_meshes))
    p = hx.frame_params(collect_counters=True)
    a = hx.DeviceScene(with_trees)
    want = a.render(p)
    want_counters = a.counters()
    b = hx.DeviceScene(without, accel=hx.ACCEL_REFERENCE_DEVICE_BUILD)
    got = b.render(p)
    assert np.array_equal(got, want)
    assert b.counters() == want_counters
    for mi in range(d.num_meshes):
        assert b.tree_stats(mi) == a.tree_stats(mi)
    # the SAH rebuild reads triangles only, so it accepts the tree-less descriptor as well
    c = hx.DeviceScene(without, accel=hx.ACCEL_SAH)
    rms = np.sqrt(np.mean((c.render(p).astype(np.float64) - want) ** 2, axis=(0, 1)))
    assert rms.max() <= 1e-4, rms


def test_device_kd_build_argument_checks(gpu):
    hx = gpu
    hs = hx.HostScene(ALL_MESHES)
    lib = hx.hip_lib()
    assert lib.hxr_kd_build_device(hs.desc, 99, None, 0, None, 0, None, None, None, None) < 0
    n, l = ctypes.c_int64(), ctypes.c_int64()
    assert lib.hxr_kd_build_device(hs.desc, 2, None, 0, None, 0, ctypes.byref(n), ctypes.byref(l), None, None) == 0 and n.value > 1
    small = np.zeros(1, dtype=hx.KD_NODE_DTYPE)
    leaves = np.zeros(1, dtype=np.int32)
    assert lib.hxr_kd_build_device(hs.desc, 2, small.ctypes.data, 1, leaves.ctypes.data, 1, None, None, None, None) < 0


def test_triangles_prepared_on_the_gpu_equal_the_hosts(gpu):
    """Mesh::prepareTriangles as a kernel (hxr_scene_options.device_prepare_triangles): gnormal, AB, AC, ABcrossAC, dNdx, dNdy of every
    triangle, bit for bit what the host loader computes (signed zeros included) -- for every reference mesh that is not recentred
    (a recentred mesh's vertices move after the reference has derived these fields from them)."""
    hx = gpu
    scene = os.path.join(ROOT, "tests", "golden", "meshes_not_recentred.hexray")
    full = hx.HostScene(scene)
    bare = hx.HostScene(scene, gpu_prepare_triangles=True, gpu_kd_build=True)
    d = SceneDesc.from_address(bare.desc)
    assert d.num_triangles == 18031
    assert all(d.triangles[k].AB[c] == 0 and d.triangles[k].gnormal[c] == 0 for k in (0, 7, d.num_triangles - 1) for c in range(3))   # nothing prepared on the host
    ds = hx.DeviceScene(bare, accel=hx.ACCEL_REFERENCE_DEVICE_BUILD, device_prepare_triangles=True)
    assert ds.verify_tables(full) == 0          # triangle records AND the hot records gathered from them
    assert hx.DeviceScene(bare, accel=hx.ACCEL_REFERENCE_DEVICE_BUILD).verify_tables(full) > 0   # the check does see unprepared records


def test_frame_from_triangles_prepared_on_the_gpu(gpu):
    """meshes.hexray through a descriptor with neither trees nor prepared triangles.  Its teapot is recentred, so the re-derived edge
    vectors differ from the reference's in the last bits: same rays, an image well inside the 1e-4 bar."""
    hx = gpu
    ov = hx.overlay_text(width=480, height=270, dof=True, num_samples=4)
    want_scene = hx.DeviceScene(hx.HostScene(scene_path("meshes.hexray"), ov))
    p = hx.frame_params(collect_counters=True)
    want = want_scene.render(p)
    bare = hx.HostScene(scene_path("meshes.hexray"), ov, gpu_prepare_triangles=True, gpu_kd_build=True)
    ds = hx.DeviceScene(bare, accel=hx.ACCEL_REFERENCE_DEVICE_BUILD, device_prepare_triangles=True)
    got = ds.render(p)
    rms = np.sqrt(np.mean((got.astype(np.float64) - want) ** 2, axis=(0, 1)))
    assert rms.max() <= 1e-4, rms
    a, b = ds.counters(), want_scene.counters()
    assert a["camera_samples"] == b["camera_samples"] and abs(a["raycasts"] - b["raycasts"]) <= 1e-4 * b["raycasts"]


def test_heightfield_prepared_on_the_gpu_from_the_raw_image(gpu):
    """hxr_heightfield::raw_heights: Gaussian blur, bounding box, normals and maxima tables of Heightfield::fillProperties as kernels.
    The blur adds the same taps in another order than the host loader (whose order is its compiler's vectorisation), so the tables
    agree to float rounding -- hxr_scene_verify_tables' tolerance -- and the frame stays well inside the 1e-4 bar."""
    hx = gpu
    ov = hx.overlay_text(width=480, height=270, dof=True, num_samples=4)
    full = hx.HostScene(scene_path("heightfield.hexray"), ov)
    bare = hx.HostScene(scene_path("heightfield.hexray"), ov, gpu_heightfield_blur=True)
    h = SceneDesc.from_address(bare.desc).heightfields[0]
    assert h.raw_heights == 1 and h.blur_radius == 8 and h.first_normal == -1 and h.first_maxh == -1 and h.first_highmap == -1
    ds = hx.DeviceScene(bare)
    assert ds.verify_tables(full) == 0
    p = hx.frame_params(collect_counters=True)
    got = ds.render(p)
    ref = hx.DeviceScene(full)
    want = ref.render(p)
    rms = np.sqrt(np.mean((got.astype(np.float64) - want) ** 2, axis=(0, 1)))
    assert rms.max() <= 1e-4, rms
    a, b = ds.counters(), ref.counters()
    assert a["camera_samples"] == b["camera_samples"]
    for k in ("raycasts", "hf_calls", "hf_tri_tests"):
        assert abs(a[k] - b[k]) <= 2e-3 * b[k], (k, a[k], b[k])


def test_device_sah_build_equals_host_sah_build_for_every_reference_mesh(gpu):
    """The module's SAH tree (hxr_scene_options.accel = HXR_ACCEL_SAH) built by kernels (hxr_sah_build_device: every candidate plane's cost
    counted independently, chunks of 256 candidates spread over the GPU) is the tree the host builder makes by sorting and sweeping
    (hxr_sah_build): same split planes bit for bit, same leaf contents in the same order, same numbering -- for all nine reference meshes."""
    hx = gpu
    hs = hx.HostScene(ALL_MESHES)
    desc = SceneDesc.from_address(hs.desc)
    total_ms = 0.0
    for mi in range(desc.num_meshes):
        want_nodes, want_leaf, want_depth, _ = hx.sah_build(hs, mi, device=False)
        nodes, leaf, depth, ms = hx.sah_build(hs, mi, device=True)
        total_ms += ms
        assert len(nodes) == len(want_nodes) and len(leaf) == len(want_leaf), (mi, len(nodes), len(want_nodes), len(leaf), len(want_leaf))
        for field in ("axis", "first", "count", "parent"):
            assert np.array_equal(nodes[field], want_nodes[field]), (mi, field)
        assert np.array_equal(nodes["split"].view(np.uint64), want_nodes["split"].view(np.uint64)), mi
        assert np.array_equal(leaf, want_leaf), mi
        assert depth == want_depth
    print("SAH trees of the nine reference meshes built on the device in %.2f ms" % total_ms)


def test_sah_scene_built_on_the_device_renders_what_the_host_built_sah_scene_renders(gpu, monkeypatch):
    """hxr_scene_create_ex with HXR_ACCEL_SAH builds its trees on the device by default; HXR_SAH_HOST=1 is the host builder: same trees,
    so bit-identical frames and identical work counters."""
    hx = gpu
    ov = hx.overlay_text(width=480, height=270, dof=True, num_samples=4)
    hs = hx.HostScene(scene_path("meshes.hexray"), ov)
    p = hx.frame_params(collect_counters=True)
    dev = hx.DeviceScene(hs, accel=hx.ACCEL_SAH)
    got = dev.render(p)
    gc = dev.counters()
    monkeypatch.setenv("HXR_SAH_HOST", "1")
    host = hx.DeviceScene(hs, accel=hx.ACCEL_SAH)
    want = host.render(p)
    wc = host.counters()
    assert np.array_equal(got, want)
    assert gc == wc
    for mi in range(3):
        assert dev.tree_stats(mi) == host.tree_stats(mi)


@pytest.mark.parametrize("name", ["meshes.hexray", "boxed.hexray", "bumpmap.hexray"])
def test_bitmaps_decoded_and_differentiated_on_the_gpu(gpu, name):
    """hxr_bitmap::prep: Bitmap::decompressGamma (BitmapTexture / cubemap assumedGamma) and Bitmap::differentiate (BumpTexture) as kernels at
    hxr_scene_create (src/bitmap.cpp:304-338).  8-bit sources are decoded through the host's 256-entry table and the derivative is plain
    float arithmetic, so the tables must be the host loader's bit for bit -- and the frame the same frame."""
    hx = gpu
    ov = hx.overlay_text(width=320, height=240)
    full = hx.HostScene(scene_path(name), ov)
    bare = hx.HostScene(scene_path(name), ov, gpu_bitmap_prep=True)
    d = SceneDesc.from_address(bare.desc)
    assert any(d.bitmaps[i].prep for i in range(d.num_bitmaps))
    ds = hx.DeviceScene(bare)
    assert ds.verify_tables(full) == 0
    p = hx.frame_params()
    assert np.array_equal(ds.render(p), hx.DeviceScene(full).render(p))


def test_gamma_curve_on_the_device_for_float_texels(gpu):
    """A bitmap whose texels are not 8-bit values (an EXR source) carries no decode table (first_gamma_table = -1): the kernel evaluates
    decompress_sRGB / pow itself.  Built here from boxed.hexray's descriptors: the table reference is dropped from the prep descriptor, and
    the module's texels must then agree with the host's to float rounding (hxr_scene_verify_tables allows 1e-6 there; the table path allows 0)."""
    hx = gpu
    for name in ("boxed.hexray", "meshes.hexray"):      # gamma 1.5 (pow) and 2.2 (the sRGB curve)
        full = hx.HostScene(scene_path(name))
        bare = hx.HostScene(scene_path(name), gpu_bitmap_prep=True)
        d = SceneDesc.from_address(bare.desc)
        n = 0
        for i in range(d.num_bitmaps):
            if d.bitmaps[i].prep & 1:
                d.bitmaps[i].first_gamma_table = -1
                n += 1
        assert n
        assert hx.DeviceScene(bare).verify_tables(full) == 0
