"""GPU tests of the interactive path (reference gameloop() src/main.cpp:443-504, foveated pass 1 :284-311,
displayVFB src/sdl.cpp:79-90): foveated frames against the oracle's restatement, the per-frame entry hxr_interactive_frame (camera
record in, 32-bit display surface out), and the `hexray --interactive` host loop."""
import ctypes
import os
import subprocess

import numpy as np
import pytest

import hexray_amd as hx
from conftest import ROOT, scene_path
from oracle_binding import oracle_render

pytestmark = pytest.mark.gpu
EXACT = ("camera_samples", "raycasts", "shadow_queries", "node_tests", "bbox_tests", "kd_visits", "tri_tests", "tex_bilinear", "env_bilinear", "aa_pixels")


def rms(a, b):
    return np.sqrt(((a.astype(np.float64) - b.astype(np.float64)) ** 2).mean(axis=(0, 1)))


FOVEATED = [
    ("simple.hexray", hx.overlay_text(960, 540), dict(mode=hx.MODE_SINGLE, foveated_radius=75, vip=(480, 270))),          # the shipped interactive scene, `foveatedRadius 75`
    ("simple.hexray", hx.overlay_text(960, 540), dict(mode=hx.MODE_SINGLE, foveated_radius=75)),                          # the mouse has not moved yet: vip (-100, -100), all coarse
    ("meshes.hexray", hx.overlay_text(333, 250), dict(mode=hx.MODE_ADAPTIVE_AA, foveated_radius=90, vip=(200, 120))),     # ragged edge blocks; AA only inside the fovea
    ("meshes.hexray", hx.overlay_text(320, 240), dict(mode=hx.MODE_ADAPTIVE_AA, foveated_radius=1000, vip=(100, 100))),   # everything inside: the plain frame
    ("heightfield.hexray", hx.overlay_text(320, 180), dict(mode=hx.MODE_SINGLE, foveated_radius=40, vip=(160, 90), tile_size=16)),
]


@pytest.mark.parametrize("scene,ov,fp", FOVEATED)
def test_foveated_frames_match_oracle(gpu, scene, ov, fp):
    hs = hx.HostScene(scene_path(scene), ov)
    ds = hx.DeviceScene(hs)
    g = ds.render(hx.frame_params(collect_counters=True, **fp))
    gc = ds.counters()
    o, oc = oracle_render(hs, hx.frame_params(**fp))
    assert np.all(rms(g, o) <= 1e-4), rms(g, o)
    assert np.abs(g - o).max() < 5e-3
    for k in EXACT:
        if "heightfield" in scene and k in ("raycasts", "node_tests", "bbox_tests", "kd_visits", "tri_tests", "tex_bilinear", "env_bilinear"):
            continue
        assert gc[k] == oc[k], (k, gc[k], oc[k])
    plain = ds.render(hx.frame_params(**{k: v for k, v in fp.items() if k not in ("foveated_radius", "vip")}))
    if fp["foveated_radius"] >= 1000:
        assert np.array_equal(g, plain)
    else:
        assert gc["camera_samples"] < g.shape[0] * g.shape[1] and not np.array_equal(g, plain)    # far fewer rays than pixels outside the fovea
    ds.close()


def test_interactive_frames_follow_the_camera(gpu):
    """hxr_interactive_frame: per frame only the camera record changes; the display surface is Color::toRGB32 of the float frame
    (checked against the host library's table, which is pinned to the reference's), and the float frame is the one a plain render
    of the same camera gives."""
    hs = hx.HostScene(scene_path("simple.hexray"), hx.overlay_text(320, 180))
    ds = hx.DeviceScene(hs)
    host = hx.host_lib()
    p = hx.frame_params(tile_size=16)
    previous = None
    for frame_no in range(3):
        hs.move_camera(0.5, 1.0)
        hs.rotate_camera(3.0, -1.0)
        surface, ms = ds.interactive_frame(p)
        f = ds.interactive_float_frame(p)
        assert ms > 0
        o, _ = oracle_render(hs, p)
        assert np.all(rms(f, o) <= 1e-4)
        fresh = hx.DeviceScene(hs)
        assert np.array_equal(f, fresh.render(p))
        fresh.close()
        lut = np.array([host.hxh_srgb8(ctypes.c_float(float(v))) for v in f.reshape(-1)], dtype=np.uint32).reshape(f.shape)
        want = (lut[..., 0] << 16) | (lut[..., 1] << 8) | lut[..., 2]
        assert np.array_equal(surface, want)
        bgr, _ = ds.interactive_frame(p, update_camera=False, shifts=(0, 8, 16))          # another surface format, same camera
        assert np.array_equal(bgr, (lut[..., 2] << 16) | (lut[..., 1] << 8) | lut[..., 0])
        assert previous is None or not np.array_equal(previous, f)
        previous = f
    ds.close()


def test_interactive_latency_of_the_shipped_scene(gpu):
    """data/simple.hexray as shipped (960x540, interactive on): steady-state latency of one loop turn, plain and foveated."""
    hs = hx.HostScene(scene_path("simple.hexray"), "")
    assert (hs.settings.frame_width, hs.settings.frame_height) == (960, 540)
    ds = hx.DeviceScene(hs)
    for fp in (hx.frame_params(tile_size=16), hx.frame_params(tile_size=16, foveated_radius=75, vip=(480, 270))):
        lat = []
        for i in range(12):
            hs.rotate_camera(0.5, 0.0)
            _, ms = ds.interactive_frame(fp)
            lat.append(ms)
        steady = sorted(lat[2:])[len(lat[2:]) // 2]
        print("interactive simple.hexray 960x540 fovea %d: median latency %.3f ms (first %.3f)" % (fp.foveated_radius, steady, lat[0]))
        assert steady < 20.0                                     # the reference's 8-thread CPU loop runs this scene at a few FPS
    ds.close()


def test_hexray_interactive_loop(gpu):
    exe = os.path.join(ROOT, "hexray_amd", "bin", "hexray")
    r = subprocess.run([exe, scene_path("simple.hexray"), "--interactive", "20", "--fovea", "75"], capture_output=True, text=True, timeout=120, cwd=ROOT)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "20 frames in" in r.stdout and "FPS" in r.stdout and "frame latency" in r.stdout and "Exited cleanly" in r.stdout


PREPASS = [
    ("meshes.hexray", hx.overlay_text(333, 250), 5),          # the default prepassSamples; ragged blocks at the right and bottom edges
    ("smallpt.hexray", hx.overlay_text(200, 150), 3),         # gi on: traceSingleRay path-traces
    ("meshes.hexray", hx.overlay_text(120, 96, dof=True, num_samples=4), 70),   # more samples per block than lanes in a wave; lens sampling
]


@pytest.mark.parametrize("scene,ov,samples", PREPASS)
def test_coarse_prepass_matches_oracle(gpu, scene, ov, samples):
    """coarseRender (src/main.cpp:387-414): every 24x24 block filled with the mean of `prepassSamples` rays through random points of it."""
    hs = hx.HostScene(scene_path(scene), ov)
    ds = hx.DeviceScene(hs)
    fp = dict(mode=hx.MODE_PREPASS, spp=samples)
    g = ds.render(hx.frame_params(collect_counters=True, **fp))
    gc = ds.counters()
    o, oc = oracle_render(hs, hx.frame_params(**fp))
    assert np.all(rms(g, o) <= 1e-4), rms(g, o)
    h, w = g.shape[:2]
    assert gc["camera_samples"] == oc["camera_samples"] == ((w + 23) // 24) * ((h + 23) // 24) * samples
    for by in range(0, h, 24):                                  # blocks are flat
        for bx in range(0, w, 24):
            blk = g[by:by + 24, bx:bx + 24]
            assert np.all(blk == blk[0, 0])
    with pytest.raises(hx.HipError):
        ds.render(hx.frame_params(mode=hx.MODE_PREPASS))       # spp is mandatory
    ds.close()
