"""Host side (loader / flattener / output) on the CPU: parser behaviour, asset decoding, kd-tree builder
statistics recorded from the reference, bucket order, BMP output."""
import os
import struct

import numpy as np
import pytest

import hexray_amd as hx
from conftest import scene_path
from desc_types import desc_of

# "nodes / leaf nodes / max depth / avg depth / avg tris per leaf" printed by the reference's
# Mesh::beginRender (reference src/mesh.cpp:39-47) for the config scenes, as recorded from the
# reference binary in SURVEY.md section 8c(iv).
REFERENCE_KD_STATS = {
    "simple.hexray": [(15, 8, 3, None, None), (6003, 3002, 65, 18.0, 11.2), (3431, 1716, 22, 12.6, 11.5)],
    "meshes.hexray": [(15, 8, 3, None, None), (1467, 734, 65, 27.6, 7.7), (6007, 3004, 65, 18.0, 11.2)],
}


@pytest.mark.parametrize("scene", sorted(REFERENCE_KD_STATS))
def test_kd_builder_matches_reference_statistics(scene):
    hs = hx.HostScene(scene_path(scene))
    got = hs.kd_stats()
    want = REFERENCE_KD_STATS[scene]
    assert len(got) == len(want)
    for g, w in zip(got, want):
        assert g[:3] == w[:3]
        if w[3] is not None:
            assert round(g[3], 1) == w[3] and round(g[4], 1) == w[4]


def test_kd_tree_structure_is_consistent():
    hs = hx.HostScene(scene_path("meshes.hexray"))
    d = desc_of(hs)
    for mi in range(d.num_meshes):
        m = d.meshes[mi]
        seen_leaf_entries = 0
        for ni in range(m.num_kd_nodes):
            n = d.kd_nodes[m.first_kd_node + ni]
            if n.axis == 3:
                assert 0 <= n.first and n.first + n.count <= m.num_leaf_indices
                seen_leaf_entries += n.count
                for k in range(n.count):
                    assert 0 <= d.leaf_indices[m.first_leaf_index + n.first + k] < m.num_triangles
            else:
                assert 0 <= n.axis <= 2 and n.first + 1 < m.num_kd_nodes
                for c in (n.first, n.first + 1):
                    assert d.kd_nodes[m.first_kd_node + c].parent == ni
                assert m.bbox_min[n.axis] <= n.split <= m.bbox_max[n.axis]
        assert seen_leaf_entries == m.num_leaf_indices
        assert d.kd_nodes[m.first_kd_node].parent == -1


def test_all_shipped_scenes_load():
    """The reference's only test is `for fn in data/*.hexray; do hexray $fn` (scripts/test_all_scenes.sh:5)."""
    import glob
    names = sorted(glob.glob(os.path.join(scene_path(""), "*.hexray")) + glob.glob(os.path.join(scene_path("hw10"), "*.hexray"))
                   + glob.glob(os.path.join(scene_path("hw12"), "*.hexray")))
    assert len(names) >= 13
    for n in names:
        hs = hx.HostScene(n)
        assert desc_of(hs).num_nodes >= 1, n


def test_settings_camera_defaults_and_overlay():
    hs = hx.HostScene(scene_path("meshes.hexray"))
    s, c = hs.settings, hs.camera
    assert (s.frame_width, s.frame_height, s.max_trace_depth, s.want_aa, s.gi) == (640, 480, 7, 1, 0)
    assert c.dof == 0 and c.num_samples == 20   # "numSamples 20 // change to 50 ..." -- trailing comment stripped
    assert abs(c.aperture_size - 2.5 / 5.6) < 1e-15 and c.focal_plane_dist == 21.48
    hs2 = hx.HostScene(scene_path("meshes.hexray"), hx.overlay_text(1920, 1080, want_aa=False, dof=True, num_samples=16))
    s, c = hs2.settings, hs2.camera
    assert (s.frame_width, s.frame_height, s.want_aa, c.dof, c.num_samples) == (1920, 1080, 0, 1, 16)


def test_camera_basis_is_orthonormal_and_matches_fov():
    hs = hx.HostScene(scene_path("simple.hexray"))
    c = hs.camera
    up, right, front = np.array(c.up), np.array(c.right), np.array(c.front)
    for v in (up, right, front):
        assert abs(np.linalg.norm(v) - 1) < 1e-12
    assert abs(up @ right) < 1e-12 and np.allclose(np.cross(right, up), front, atol=1e-12)
    # the field of view is matched on the frame DIAGONAL (reference src/camera.cpp:39-41)
    pos, tl, tr, bl = (np.array(x) for x in (c.pos, c.top_left, c.top_right, c.bottom_left))
    centre = (tr + bl) / 2
    half_diag = np.linalg.norm(tl - centre) / np.linalg.norm(centre - pos)
    assert abs(np.degrees(np.arctan(half_diag)) * 2 - 100) < 1e-9


def test_parser_errors_are_reported():
    with pytest.raises(ValueError, match="Unknown object class"):
        hx.HostScene(scene_path("simple.hexray"), "Torus t {\n}\n")
    with pytest.raises(ValueError, match="outside the allowed bounds"):
        hx.HostScene(scene_path("simple.hexray"), "Camera {\n fov 500\n}\n")
    with pytest.raises(ValueError, match="not defined"):
        hx.HostScene(scene_path("simple.hexray"), "Node extra {\n geometry nothing\n shader flat\n}\n")
    with pytest.raises(ValueError):
        hx.HostScene(scene_path("does_not_exist.hexray"))


def test_parser_comment_and_quote_rules(tmp_path):
    text = """GlobalSettings {
	frameWidth 32   // trailing comment
	frameHeight 24  # hash comment
}
/* block comment
Camera bogus { }
*/
Camera cam {
	pos (1, 2, 3)
}
Plane floor {
}
Lambert l {
}
Node n {
	geometry floor
	shader l
	translate (1, 0, 0)
	scale (2, 2, 2)
}
"""
    p = tmp_path / "t.hexray"
    p.write_text(text)
    hs = hx.HostScene(str(p))
    assert (hs.settings.frame_width, hs.settings.frame_height) == (32, 24)
    assert list(hs.camera.pos) == [1.0, 2.0, 3.0]
    T = desc_of(hs).nodes[0].T
    assert list(T.offset) == [1.0, 0.0, 0.0] and T.m[0] == 2.0 and T.inv_m[0] == 0.5


def test_bmp_decoder_against_pillow():
    from PIL import Image
    for rel in ("texture/zar-texture.bmp", "texture/heightfield.bmp", "env/mountain/posx.bmp"):
        ours = hx.load_image(scene_path(rel))
        ref = np.asarray(Image.open(scene_path(rel)).convert("RGB"), dtype=np.float32) / np.float32(255.0)
        assert ours.shape == ref.shape
        # the host library is built with the reference's Release flags (-ffast-math turns `/ 255.0f` into a
        # reciprocal multiply), so texel values may differ from the exact quotient by one ulp
        assert np.array_equal(np.round(ours * 255).astype(np.uint8), np.round(ref * 255).astype(np.uint8))
        assert np.allclose(ours, ref, rtol=2.5e-7, atol=0)


def test_bucket_list_is_the_reference_zigzag():
    b = hx.bucket_list(1920, 1080, 64)
    assert len(b) == 30 * 17   # 510 buckets at 1080p
    assert tuple(b[0]) == (0, 0, 64, 64) and tuple(b[29]) == (1856, 0, 1920, 64)
    assert tuple(b[30]) == (1856, 64, 1920, 128)          # odd rows run right to left
    assert tuple(b[-1])[3] == 1080 and all(r[2] <= 1920 and r[3] <= 1080 for r in b)
    covered = np.zeros((1080, 1920), dtype=np.int32)
    for x0, y0, x1, y1 in b:
        covered[y0:y1, x0:x1] += 1
    assert covered.min() == 1 and covered.max() == 1


def test_bmp_writer_layout(tmp_path):
    frame = np.zeros((3, 5, 3), dtype=np.float32)
    frame[0, 0] = (1.0, 0.0, 0.0)
    frame[2, 4] = (0.0, 0.5, 2.0)
    path = str(tmp_path / "o.bmp")
    hx.save_bmp(path, frame)
    raw = open(path, "rb").read()
    assert raw[:2] == b"BM" and struct.unpack_from("<iii", raw, 2)[2] == 54
    w, h = struct.unpack_from("<ii", raw, 18)
    assert (w, h) == (5, 3) and len(raw) == 54 + 16 * 3   # rows padded to 4 bytes
    from PIL import Image
    px = np.asarray(Image.open(path).convert("RGB"))
    assert tuple(px[0, 0]) == (255, 0, 0)
    lib = hx.host_lib()
    import ctypes
    assert tuple(px[2, 4]) == (0, lib.hxh_srgb8(ctypes.c_float(0.5)), 255)


def test_heightfield_tables():
    hs = hx.HostScene(scene_path("heightfield.hexray"))
    d = desc_of(hs)
    f = d.heightfields[0]
    assert (f.W, f.H, f.max_k, f.use_optimization) == (512, 512, 9, 1)
    n = f.W * f.H
    heights = np.ctypeslib.as_array(d.f32_pool, shape=(d.f32_pool_size,))[f.first_height:f.first_height + n].reshape(f.H, f.W)
    maxh = np.ctypeslib.as_array(d.f32_pool, shape=(d.f32_pool_size,))[f.first_maxh:f.first_maxh + n].reshape(f.H, f.W)
    high = np.ctypeslib.as_array(d.f32_pool, shape=(d.f32_pool_size,))[f.first_highmap:f.first_highmap + n * 16].reshape(f.H, f.W, 16)
    # maxH = max over the cell's 2x2 corners (reference src/heightfield.cpp:223-233)
    pad = np.pad(heights, ((0, 1), (0, 1)), mode="edge")
    want = np.maximum(np.maximum(pad[:-1, :-1], pad[:-1, 1:]), np.maximum(pad[1:, :-1], pad[1:, 1:]))
    assert np.array_equal(maxh, want)
    # the pyramid is monotone in k and bounds the heights (reference src/heightfield.cpp:49-81)
    assert np.all(high[:, :, 0] >= heights)
    for k in range(1, f.max_k):
        assert np.all(high[:, :, k] >= high[:, :, k - 1] - 0)
    assert f.bbox_min[1] == heights.min() and f.bbox_max[1] == heights.max()


def test_exr_piz_decoder_yields_consistent_cube_maps():
    """The EXR decoder is unpinned against OpenEXR (absent from the image); what can be checked: every file decodes
    completely (each chunk to its exact size, Huffman streams ending on their boundaries -- otherwise load fails),
    texels are finite HDR radiance, and neighbouring cube faces continue each other across their shared edge."""
    for env in ("forest", "ocean"):
        faces = {n: hx.load_image(scene_path(f"env/{env}/{n}.exr")) for n in ("negx", "negy", "negz", "posx", "posy", "posz")}
        for img in faces.values():
            assert img.shape == (256, 256, 3) and np.isfinite(img).all() and img.min() >= 0 and img.max() > 0.5
        # horizontal ring negx | posz | posx | negz: last column of one face ~ first column of the next; the negz face
        # is stored turned by 180 degrees (it is looked up with (x, +y) where posz uses (x, -y), reference
        # src/environment.cpp:73-78)
        faces["negz"] = faces["negz"][::-1, ::-1]
        ring = ["negx", "posz", "posx", "negz", "negx"]
        for a, b in zip(ring[:-1], ring[1:]):
            ea, eb = np.log1p(faces[a][:, -1]), np.log1p(faces[b][:, 0])
            inner = np.log1p(faces[a][:, -2])
            seam = np.abs(ea - eb).mean()
            step = np.abs(ea - inner).mean()
            assert seam < 4 * step + 0.02, (env, a, b, seam, step)
    hs = hx.HostScene(scene_path("simple.hexray"))
    assert not any("could not load maps" in w for w in hs.warnings)
    assert desc_of(hs).environment.loaded == 1


def test_exr_writer_round_trip(tmp_path):
    """Bitmap::saveEXR's replacement (RGBA HALF, alpha 1, written uncompressed): what it writes, the EXR reader reads back, and the
    texels are the float frame rounded to 16-bit HALF exactly as numpy's float16 conversion (round to nearest even) does."""
    import ctypes
    rng = np.random.default_rng(5)
    frame = (rng.random((37, 53, 3), dtype=np.float32) * np.float32(4.0)).astype(np.float32)
    frame[0, 0] = (0.0, 1.0, 65504.0)            # zero, one, the largest HALF
    frame[0, 1] = (1e-7, 6.1e-5, 70000.0)        # subnormal HALF range, overflow to inf
    frame[0, 2] = (-0.25, 0.333333, 2049.0)      # sign, rounding, a tie (2049 -> 2048: even)
    path = str(tmp_path / "shot.exr")
    hx.save_exr(path, frame)
    back = hx.load_image(path)
    assert back.shape == frame.shape
    want = frame.astype(np.float16).astype(np.float32)
    assert np.array_equal(back, want)
    lib = hx.host_lib()
    for v in (0.0, 1.0, -2.5, 1e-7, 6.1e-5, 65504.0, 65520.0, 2049.0, 2051.0, 0.1):
        assert lib.hxh_float_to_half(ctypes.c_float(v)) == int(np.float32(v).astype(np.float16).view(np.uint16)), v
    assert open(path, "rb").read(4) == bytes([0x76, 0x2f, 0x31, 0x01])


def test_loader_can_leave_the_kd_trees_to_the_render_module():
    """HXH_GPU_KD_BUILD: meshes keep has_kd but carry no nodes / leaf lists; everything else of the descriptor is unchanged."""
    import hexray_amd as hx
    from desc_types import SceneDesc
    full = hx.HostScene(scene_path("meshes.hexray"))
    bare = hx.HostScene(scene_path("meshes.hexray"), gpu_kd_build=True)
    a, b = SceneDesc.from_address(full.desc), SceneDesc.from_address(bare.desc)
    assert a.num_kd_nodes > 0 and b.num_kd_nodes == 0 and b.num_leaf_indices == 0
    assert a.num_meshes == b.num_meshes and a.num_triangles == b.num_triangles and a.f64_pool_size == b.f64_pool_size
    for i in range(a.num_meshes):
        assert b.meshes[i].has_kd == a.meshes[i].has_kd and b.meshes[i].num_kd_nodes == 0
        assert list(b.meshes[i].bbox_min) == list(a.meshes[i].bbox_min) and list(b.meshes[i].bbox_max) == list(a.meshes[i].bbox_max)


def test_heightfield_blur_below_one_sixth_flattens_the_field_on_both_paths():
    """0 < blur < 1/6: nearestInt(3 * blur) is 0, the reference's tap loops (src/heightfield.cpp:204-212) are empty and every height becomes
    0.  The host-prepared tables reproduce that; the raw-image descriptor (blur left to the module, where a radius of 0 means "no blur")
    must hand over the same flat field."""
    ov = "Heightfield island {\n\tblur 0.1\n}\n"
    full = hx.HostScene(scene_path("heightfield.hexray"), ov)
    raw = hx.HostScene(scene_path("heightfield.hexray"), ov, gpu_heightfield_blur=True)
    for hs in (full, raw):
        d = desc_of(hs)
        f = d.heightfields[0]
        n = f.W * f.H
        heights = np.ctypeslib.as_array(d.f32_pool, shape=(d.f32_pool_size,))[f.first_height:f.first_height + n]
        assert not heights.any()
    assert desc_of(raw).heightfields[0].raw_heights == 1 and desc_of(raw).heightfields[0].blur_radius == 0


def test_bitmap_passes_left_to_the_module():
    """HXH_GPU_BITMAP_PREP: bitmaps are handed over as their files hold them, flagged with the passes still due (hxr_bitmap::prep) and, for 8-bit
    sources, the 256-entry decode table; the rest of the descriptor is unchanged.  Applying the passes here (numpy) gives the complete
    descriptor's texels bit for bit: gamma by table look-up, Bitmap::differentiate (src/bitmap.cpp:304-319) with the Release build's
    intensity = (r + g + b) * (1/3)."""
    seen = set()
    for name in ("meshes.hexray", "boxed.hexray", "bumpmap.hexray"):
        full_scene, bare_scene = hx.HostScene(scene_path(name)), hx.HostScene(scene_path(name), gpu_bitmap_prep=True)
        full, bare = desc_of(full_scene), desc_of(bare_scene)
        assert full.num_bitmaps == bare.num_bitmaps and full.num_textures == bare.num_textures and full.num_triangles == bare.num_triangles
        fpool = np.ctypeslib.as_array(full.f32_pool, shape=(full.f32_pool_size,))
        bpool = np.ctypeslib.as_array(bare.f32_pool, shape=(bare.f32_pool_size,))
        for i in range(full.num_bitmaps):
            f, b = full.bitmaps[i], bare.bitmaps[i]
            assert f.prep == 0 and (f.width, f.height) == (b.width, b.height)
            n = f.width * f.height * 3
            want = fpool[f.first_texel:f.first_texel + n]
            t = bpool[b.first_texel:b.first_texel + n].copy()
            seen.add(b.prep)
            if b.prep & 1:
                assert b.first_gamma_table >= 0      # the shipped textures are 8-bit BMPs
                table = bpool[b.first_gamma_table:b.first_gamma_table + 256]
                q = (t * np.float32(255.0) + np.float32(0.5)).astype(np.int32)
                assert q.min() >= 0 and q.max() <= 255 and np.abs(q / 255.0 - t).max() < 1e-7
                t = np.where(t > 0, table[q], t)
            if b.prep & 2:
                img = t.reshape(f.height, f.width, 3)
                inten = ((img[..., 0] + img[..., 1]) + img[..., 2]) * (np.float32(1.0) / np.float32(3.0))
                out = np.zeros_like(img)
                out[..., 0] = inten - np.roll(inten, -1, axis=1)
                out[..., 1] = inten - np.roll(inten, -1, axis=0)
                t = out.reshape(-1)
            assert np.array_equal(t.view(np.uint32), want.view(np.uint32)), (name, i, b.prep)
    assert {1, 2} <= seen     # a gamma-decoded texture / cubemap and a BumpTexture were among them
