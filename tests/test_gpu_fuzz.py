"""Differential test: randomly generated scenes (analytic primitives, CSG one and two operations deep, transformed nodes, every shader
and light class, Whitted / adaptive AA / depth of field / path tracing) through the HIP module and through the CPU oracle -- two
independent implementations of the same restatement.  Per-channel RMS <= 1e-4 and identical ray / shadow / node counters."""
import os
import random

import numpy as np
import pytest

import hexray_amd as hx
from conftest import ROOT
from oracle_binding import oracle_render

pytestmark = pytest.mark.gpu


def v3(r, lo, hi):
    return "(%.4f, %.4f, %.4f)" % tuple(r.uniform(lo, hi) for _ in range(3))


def col(r, lo=0.1, hi=1.0):
    return v3(r, lo, hi)


def make_scene(seed, geom_dir=None, without=()):
    """geom_dir: path of data/geom relative to where the scene file will live (asset paths are relative to the scene file)."""
    r = random.Random(seed)
    data_dir = os.path.dirname(geom_dir) if geom_dir else None
    mode = ("whitted", "aa", "dof", "gi")[seed % 4]
    t = []
    t.append("GlobalSettings {\n\tframeWidth 80\n\tframeHeight 60\n\tambientLight %s\n\twantAA %s\n\tprepassSamples 0\n\tmaxTraceDepth %d\n%s}\n"
             % (col(r, 0.05, 0.3), "on" if mode == "aa" else "off", r.randint(2, 5), "\tgi on\n\tnumPaths 3\n" if mode == "gi" else ""))
    t.append("Camera camera {\n\tpos (%.3f, %.3f, %.3f)\n\taspectRatio 1.33333\n\tyaw %.2f\n\tpitch %.2f\n\troll %.2f\n\tfov %.1f\n%s}\n"
             % (r.uniform(-8, 8), r.uniform(10, 30), r.uniform(-60, -40), r.uniform(-10, 10), r.uniform(-25, -5), r.uniform(-5, 5), r.uniform(50, 90),
                ("\tdof on\n\tfNumber %.1f\n\tfocalPlaneDist %.2f\n\tnumSamples 3\n%s" % (r.uniform(2, 8), r.uniform(40, 60), "\tautoFocus on\n" if seed % 8 == 6 else "") if mode == "dof" else "")
                + ("\tstereoSeparation %.2f\n" % r.uniform(0.5, 3) if seed % 7 == 3 else "")))
    for i in range(r.randint(1, 2)):
        t.append("PointLight pl%d {\n\tpos (%.2f, %.2f, %.2f)\n\tpower %.0f\n\tcolor %s\n}\n" % (i, r.uniform(-60, 60), r.uniform(50, 120), r.uniform(-80, 20), r.uniform(15000, 60000), col(r, 0.5, 1)))
    if mode == "gi" or r.random() < 0.3:
        t.append("RectLight rl {\n\ttranslate (%.2f, %.2f, %.2f)\n\trotate (%.1f, %.1f, 0)\n\tscale (14, 14, 14)\n\tcolor (1, 1, 1)\n\tpower %.0f\n\txSubd 2\n\tySubd 2\n}\n"
                 % (r.uniform(-15, 15), r.uniform(45, 70), r.uniform(-15, 15), r.uniform(-20, 20), r.uniform(-20, 20), r.uniform(3000, 9000)))
    # shaders
    t.append("CheckerTexture chk {\n\tcolor1 %s\n\tcolor2 %s\n\tscaling %.2f\n}\n" % (col(r), col(r, 0, 0.4), r.uniform(3, 12)))
    t.append("Lambert lam_tex {\n\tcolor (1, 1, 1)\n\ttexture chk\n}\n")
    t.append("Lambert lam {\n\tcolor %s\n}\n" % col(r))
    t.append("Phong pho {\n\tcolor %s\n\texponent %.1f\n}\n" % (col(r), r.uniform(5, 120)))
    glossy = mode in ("whitted", "dof") and r.random() < 0.4
    t.append("Reflection refl {\n\treflColor %s\n%s}\n" % (col(r, 0.5, 1), "\tglossiness %.2f\n\tnumSamples 2\n" % r.uniform(0.3, 0.9) if glossy else ""))
    t.append("Refraction refr {\n\tior %.3f\n\trefrColor %s\n}\n" % (r.uniform(1.1, 1.8), col(r, 0.6, 1)))
    t.append("Fresnel fres {\n\tior %.3f\n}\n" % r.uniform(1.2, 1.7))
    t.append("Layered glass {\n\tlayer refr (1, 1, 1)\n\tlayer refl (1, 1, 1) fres\n}\n")
    t.append("Layered shiny {\n\tlayer pho (1, 1, 1)\n\tlayer refl (%.2f, %.2f, %.2f)\n}\n" % ((r.uniform(0.1, 0.5),) * 3))
    shaders = ["lam_tex", "lam", "pho", "shiny", "glass", "refl"] if mode != "gi" else ["lam_tex", "lam", "refl", "refr", "pho"]
    bumps = [None]
    if data_dir and "bitmaps" not in without:   # bitmap-backed lookups: texture, bump map, procedural bumps, cubemap environment
        t.append("BitmapTexture wood {\n\tfile \"%s/texture/wood.bmp\"\n\tscaling %.2f\n}\nLambert lam_wood {\n\tcolor (1, 1, 1)\n\ttexture wood\n}\n" % (data_dir, r.uniform(5, 40)))
        t.append("BumpTexture bump_tex {\n\tfile \"%s/texture/zar-bump.bmp\"\n\tstrength %.2f\n}\nBumps bump_proc {\n\tstrength %.2f\n}\n" % (data_dir, r.uniform(1, 6), r.uniform(0.05, 0.3)))
        if "wood" not in without:
            shaders.append("lam_wood")
        if "bump" not in without:
            bumps += ["bump_tex", "bump_proc"]
        if seed % 4 in (0, 2) and seed % 8 != 0 and "env" not in without:
            t.append("CubemapEnvironment {\n\tfolder \"%s/env/%s\"\n}\n" % (data_dir, r.choice(("forest", "ocean"))))
    # ground
    t.append("Plane ground {\n\ty 0\n\tlimit %.0f\n}\nNode ground_node {\n\tgeometry ground\n\tshader lam_tex\n}\n" % r.uniform(80, 400))
    # primitives and CSG
    geoms = []
    for i in range(r.randint(3, 6)):
        kind = r.choice(("Sphere", "Cube", "Sphere", "Cube", "Plane"))
        name = "g%d" % i
        if kind == "Sphere":
            t.append("Sphere %s {\n\tO %s\n\tR %.3f\n}\n" % (name, v3(r, -2, 2), r.uniform(3, 8)))
        elif kind == "Cube":
            t.append("Cube %s {\n\tO %s\n\tside %.3f\n}\n" % (name, v3(r, -2, 2), r.uniform(5, 12)))
        else:
            t.append("Plane %s {\n\ty %.3f\n\tlimit %.2f\n}\n" % (name, r.uniform(-2, 2), r.uniform(4, 10)))
        geoms.append((name, kind))
    solids = [g for g, k in geoms if k != "Plane"]
    csgs = []
    inside = {g: {g} for g in solids}   # the primitives below each operand: the two operands of an operation never share one (coinciding
    if len(solids) >= 2:                #  surfaces tie in CSGBase::intersect's std::sort, whose order of equal keys is unspecified)
        for i in range(r.randint(0, 2)):
            a, b = r.sample(solids + csgs, 2) if csgs and r.random() < 0.5 else r.sample(solids, 2)
            if inside[a] & inside[b]:
                continue
            name = "op%d" % i
            t.append("%s %s {\n\tleft %s\n\tright %s\n}\n" % (r.choice(("CSGUnion", "CSGInter", "CSGDiff")), name, a, b))
            csgs.append(name)
            inside[name] = inside[a] | inside[b]
    if data_dir and seed % 5 == 2 and "hf" not in without:   # a heightfield (512 x 512 texels scaled down to the scene), blurred or not
        t.append("Heightfield hf {\n\tfile \"%s/texture/heightfield.bmp\"\n\tblur %.1f\n\tuseOptimization %s\n}\n" % (data_dir, r.choice((0.0, 1.5, 2.5)), r.choice(("true", "false"))))
        t.append("Node n_hf {\n\tgeometry hf\n\tshader %s\n\tscale (0.12, %.1f, 0.12)\n\ttranslate (%.1f, 0.5, %.1f)\n}\n" % (r.choice(("lam", "lam_tex", "pho")), r.uniform(8, 25), r.uniform(-40, -20), r.uniform(-10, 10)))
    if geom_dir and seed % 3 != 1:   # a kd-tree mesh among the analytic shapes (never a CSG operand: undefined in the reference)
        obj, scale = r.choice((("teapot_lowres.obj", 6.0), ("heart.obj", 5.0), ("truncated_cube.obj", 5.0)))
        t.append("Mesh mesh0 {\n\tfile \"%s/%s\"\n%s%s}\n" % (geom_dir, obj, "\tautoSmooth true\n" if obj == "heart.obj" else "", "\tfaceted true\n" if obj == "truncated_cube.obj" else ""))
        t.append("Node n_mesh0 {\n\tgeometry mesh0\n\tshader %s\n\tscale (%.2f, %.2f, %.2f)\n\trotate (%.1f, %.1f, 0)\n\ttranslate (%.2f, %.2f, %.2f)\n}\n"
                 % (r.choice(shaders), scale, scale * r.uniform(0.8, 1.2), scale, r.uniform(-90, 90), r.uniform(-20, 20), r.uniform(-15, 15), r.uniform(6, 14), r.uniform(-10, 15)))
    placed = 0
    for name in [g for g, _ in geoms] + csgs:
        if r.random() < 0.25 and placed >= 2:
            continue
        tr = "\ttranslate (%.2f, %.2f, %.2f)\n" % (r.uniform(-25, 25), r.uniform(4, 16), r.uniform(-12, 25))
        if r.random() < 0.6:
            tr = "\trotate (%.1f, %.1f, %.1f)\n" % (r.uniform(-60, 60), r.uniform(-40, 40), r.uniform(-30, 30)) + tr
        if r.random() < 0.4:
            tr = "\tscale (%.2f, %.2f, %.2f)\n" % (r.uniform(0.6, 1.6), r.uniform(0.6, 1.6), r.uniform(0.6, 1.6)) + tr
        bump = r.choice(bumps) if r.random() < 0.4 else None
        t.append("Node n_%s {\n\tgeometry %s\n\tshader %s\n%s%s}\n" % (name, name, r.choice(shaders), "\tbump %s\n" % bump if bump else "", tr))
        placed += 1
    return "".join(t), mode


@pytest.mark.parametrize("seed", range(int(os.environ.get("HXR_FUZZ_FIRST", "0")), int(os.environ.get("HXR_FUZZ_FIRST", "0")) + int(os.environ.get("HXR_FUZZ_SEEDS", "60"))))
def test_random_scene_matches_oracle(gpu, tmp_path, seed):
    text, mode = make_scene(seed, os.path.relpath(os.path.join(ROOT, "data", "geom"), tmp_path))
    path = os.path.join(tmp_path, "fuzz_%d.hexray" % seed)
    with open(path, "w") as f:
        f.write(text)
    hs = hx.HostScene(path)
    p = hx.frame_params(collect_counters=True)
    ds = hx.DeviceScene(hs)
    got = ds.render(p)
    gc = ds.counters()
    want, oc = oracle_render(hs, p)
    rms = np.sqrt(np.mean((got.astype(np.float64) - want) ** 2, axis=(0, 1)))
    assert np.all(np.isfinite(got)) and rms.max() <= 1e-4, (seed, mode, rms)
    assert gc["camera_samples"] == oc["camera_samples"]
    # The work counters are EQUAL: every ray, shadow query, node / box / kd / triangle test, texture and heightfield access of the frame.
    # tools/fuzz_mismatch.py runs the first 600 generated scenes (round 3): no counter differs in any of them.  (Rounds 1-2 tolerated
    # max(8, 0.1 %): two of 600 scenes then flipped a float comparison -- refract()'s total internal reflection, a grazed heightfield cell.)
    for k in ("raycasts", "shadow_queries", "node_tests", "bbox_tests", "kd_visits", "tri_tests", "tex_bilinear", "env_bilinear", "hf_calls", "hf_tri_tests"):
        assert gc[k] == oc[k], (seed, mode, k, gc[k], oc[k])
    ds.close()


@pytest.mark.parametrize("seed", range(2, 2 + 2 * int(os.environ.get("HXR_FUZZ_ENTRY_SEEDS", "12")), 2))
def test_random_scene_through_the_other_entry_points(gpu, tmp_path, seed):
    """The same generated scenes through what else the ABI offers, each against the plain frame: three shards assembled on the GPU, an
    odd tile size, a repeated frame (the adaptive schedule at work), deferred secondary rays, a scene whose kd-trees / heightfield tables
    were prepared by the module, a device group of two handles on the one GPU."""
    import torch
    from hexray_amd.distributed import padded_tiles_per_rank
    geom = os.path.relpath(os.path.join(ROOT, "data", "geom"), tmp_path)
    text, mode = make_scene(seed, geom)
    path = os.path.join(tmp_path, "fuzz_%d.hexray" % seed)
    with open(path, "w") as f:
        f.write(text)
    hs = hx.HostScene(path)
    ds = hx.DeviceScene(hs)
    W, H = hs.settings.frame_width, hs.settings.frame_height
    base = ds.render(hx.frame_params())
    assert np.array_equal(ds.render(hx.frame_params()), base)                     # scheduled from the first frame's cost map
    assert np.array_equal(ds.render(hx.frame_params(tile_size=24)), base)        # packets in another order
    if mode != "aa":   # (sharded adaptive AA is covered by tests/test_gpu_parity.py: it needs the halo exchange set up there)
        world, tile = 3, 16
        slots = padded_tiles_per_rank(W, H, tile, world)
        parts = []
        for rank in range(world):
            p = hx.frame_params(tile_size=tile, shard_rank=rank, shard_count=world, layout=hx.LAYOUT_TILES)
            buf = torch.zeros(slots * tile * tile * 3, dtype=torch.float32, device="cuda")
            ds.render_device(p, buf.data_ptr(), 0, torch.cuda.current_stream().cuda_stream)
            parts.append(buf)
        frame = torch.zeros((H, W, 3), dtype=torch.float32, device="cuda")
        p0 = hx.frame_params(tile_size=tile, shard_rank=0, shard_count=world, layout=hx.LAYOUT_TILES)
        ds.assemble_frame(p0, torch.cat(parts).data_ptr(), slots * tile * tile * 12, frame.data_ptr(), W * 12, torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        assert np.array_equal(frame.cpu().numpy(), base)

    def close(a, b=base):
        return np.sqrt(np.mean((a.astype(np.float64) - b) ** 2, axis=(0, 1))).max() <= 1e-4

    assert close(hx.DeviceScene(hs, deferred_rays=True).render(hx.frame_params()))           # another summation order at most
    bare = hx.HostScene(path, gpu_kd_build=True, gpu_heightfield_tables=True)
    assert np.array_equal(hx.DeviceScene(bare, accel=hx.ACCEL_REFERENCE_DEVICE_BUILD).render(hx.frame_params()), base)
    assert close(hx.DeviceScene(hs, accel=hx.ACCEL_SAH).render(hx.frame_params()))
    raw = hx.HostScene(path, gpu_heightfield_blur=True, gpu_prepare_triangles=True, gpu_kd_build=True)
    assert close(hx.DeviceScene(raw, accel=hx.ACCEL_REFERENCE_DEVICE_BUILD, device_prepare_triangles=True).render(hx.frame_params()))
    group = hx.DeviceGroup(hs, [0, 0])
    assert np.array_equal(group.render(hx.frame_params()), base)
    group.close()


@pytest.mark.parametrize("seed", (0, 4, 8, 12, 16, 20))
def test_random_scene_other_render_modes(gpu, tmp_path, seed):
    """Generated Whitted scenes through the modes render() does not pick by itself: pass 1 alone, the coarse prepass, foveated pass 1 and
    foveated adaptive AA -- against the oracle's restatements of the same loops."""
    text, mode = make_scene(seed, os.path.relpath(os.path.join(ROOT, "data", "geom"), tmp_path))
    assert mode == "whitted"
    path = os.path.join(tmp_path, "fuzz_%d.hexray" % seed)
    with open(path, "w") as f:
        f.write(text)
    hs = hx.HostScene(path)
    ds = hx.DeviceScene(hs)
    for fp in (dict(mode=hx.MODE_SINGLE), dict(mode=hx.MODE_PREPASS, spp=3), dict(mode=hx.MODE_SINGLE, foveated_radius=18, vip=(30, 25)),
               dict(mode=hx.MODE_ADAPTIVE_AA, foveated_radius=22, vip=(50, 30)), dict(mode=hx.MODE_MONTE_CARLO, spp=5), dict(mode=hx.MODE_ADAPTIVE_AA)):
        got = ds.render(hx.frame_params(collect_counters=True, **fp))
        gc = ds.counters()
        want, oc = oracle_render(hs, hx.frame_params(**fp))
        rms = np.sqrt(np.mean((got.astype(np.float64) - want) ** 2, axis=(0, 1)))
        assert rms.max() <= 1e-4, (seed, fp, rms)
        assert gc["camera_samples"] == oc["camera_samples"] and abs(gc["raycasts"] - oc["raycasts"]) <= 8, (seed, fp, gc["raycasts"], oc["raycasts"])


@pytest.mark.parametrize("seed", (1, 2, 3, 5))
def test_random_scene_under_a_moving_camera(gpu, tmp_path, seed):
    """hxr_scene_update_camera on generated scenes: after every move the frame equals the oracle's for the moved camera -- the cost map
    the adaptive schedule carries over from the previous camera position may only reorder work -- and sharded adaptive-AA frames (halo
    pixels, pass-1 schedule) still assemble to the whole frame."""
    import random as _r
    text, mode = make_scene(seed, os.path.relpath(os.path.join(ROOT, "data", "geom"), tmp_path))
    path = os.path.join(tmp_path, "fuzz_%d.hexray" % seed)
    with open(path, "w") as f:
        f.write(text)
    hs = hx.HostScene(path)
    ds = hx.DeviceScene(hs)
    r = _r.Random(seed)
    for step in range(4):
        hs.move_camera(r.uniform(-3, 3), r.uniform(-2, 4))
        hs.rotate_camera(r.uniform(-8, 8), r.uniform(-3, 3))
        ds.update_camera()
        got = ds.render(hx.frame_params())
        want, _ = oracle_render(hs, hx.frame_params())
        rms = np.sqrt(np.mean((got.astype(np.float64) - want) ** 2, axis=(0, 1)))
        assert rms.max() <= 1e-4, (seed, mode, step, rms)
    if mode == "aa":
        whole = ds.render(hx.frame_params(mode=hx.MODE_ADAPTIVE_AA))
        for count in (2, 3):
            acc = np.zeros_like(whole)
            for rank in range(count):
                part = ds.render(hx.frame_params(mode=hx.MODE_ADAPTIVE_AA, tile_size=16, shard_rank=rank, shard_count=count))
                acc += part          # LAYOUT_FRAME: a rank's frame holds its own tiles and zeros elsewhere
            assert np.array_equal(acc, whole), (seed, count)
