"""Properties of the CPU oracle's frame drivers (CPU only): schedule independence, adaptive AA
bookkeeping, Monte-Carlo convergence, sensitivity to the compiler's floating-point contraction."""
import os
import subprocess

import numpy as np
import pytest

import hexray_amd as hx
from conftest import ROOT, scene_path
from oracle_binding import oracle_lib, oracle_render


def rms(a, b):
    return np.sqrt(((a.astype(np.float64) - b.astype(np.float64)) ** 2).mean(axis=(0, 1)))


def test_thread_count_does_not_change_the_image():
    """The reference is only reproducible single-threaded in MC modes; the oracle's
    counter-based generator makes every mode independent of the schedule."""
    hs = hx.HostScene(scene_path("meshes.hexray"), hx.overlay_text(96, 72, dof=True, num_samples=4))
    a, ca = oracle_render(hs, hx.frame_params(), threads=1)
    b, cb = oracle_render(hs, hx.frame_params(), threads=5)
    assert np.array_equal(a, b) and ca == cb


def test_adaptive_aa_only_touches_flagged_pixels():
    hs = hx.HostScene(scene_path("meshes.hexray"), hx.overlay_text(160, 120))
    one, c1 = oracle_render(hs, hx.frame_params(mode=hx.MODE_SINGLE))
    aa, c2 = oracle_render(hs, hx.frame_params(mode=hx.MODE_ADAPTIVE_AA))
    changed = np.any(one != aa, axis=2)
    assert 0 < c2["aa_pixels"] < 160 * 120 * 0.5
    assert changed.sum() <= c2["aa_pixels"]
    # four extra taps per flagged pixel (reference src/main.cpp:336-338)
    assert c2["camera_samples"] == c1["camera_samples"] + 4 * c2["aa_pixels"]
    # re-derive the flags from the pass-1 frame (reference src/main.cpp:192-222)
    c = np.minimum(one, 1.0)
    flags = np.zeros(one.shape[:2], dtype=bool)
    H, W = flags.shape
    for dy in (-1, 0, 1):
        for dx in (-1, 0, 1):
            if dx == 0 and dy == 0:
                continue
            ys, xs = slice(max(0, -dy), H - max(0, dy)), slice(max(0, -dx), W - max(0, dx))
            yn, xn = slice(max(0, dy), H - max(0, -dy)), slice(max(0, dx), W - max(0, -dx))
            flags[ys, xs] |= np.any(np.abs(c[ys, xs] - c[yn, xn]) > np.float32(0.075), axis=2)
    assert flags.sum() == c2["aa_pixels"]
    assert not np.any(changed & ~flags)


def test_monte_carlo_converges():
    """More samples -> closer to a high-sample frame, roughly as 1/sqrt(spp)."""
    def frame(spp, seed):
        hs = hx.HostScene(scene_path("smallpt.hexray"), hx.overlay_text(48, 36, num_paths=spp))
        return oracle_render(hs, hx.frame_params(seed=seed))[0]
    ref = frame(1024, 1)
    e4, e64 = rms(frame(4, 7), ref).mean(), rms(frame(64, 7), ref).mean()
    assert e64 < e4 * 0.45     # 1/sqrt(16) = 0.25 expected
    assert 0.1 < ref.mean() < 2.0


def test_seed_changes_monte_carlo_but_not_deterministic_modes():
    hs = hx.HostScene(scene_path("meshes.hexray"), hx.overlay_text(64, 48, dof=True, num_samples=2))
    a = oracle_render(hs, hx.frame_params(seed=1))[0]
    b = oracle_render(hs, hx.frame_params(seed=2))[0]
    assert not np.array_equal(a, b)
    hs2 = hx.HostScene(scene_path("meshes.hexray"), hx.overlay_text(64, 48))
    c = oracle_render(hs2, hx.frame_params(mode=hx.MODE_ADAPTIVE_AA, seed=1))[0]
    d = oracle_render(hs2, hx.frame_params(mode=hx.MODE_ADAPTIVE_AA, seed=2))[0]
    assert np.array_equal(c, d)


def test_fast_math_build_matches_strict_build():
    """The reference ships with -O3 -ffast-math -march=haswell (CMakeLists.txt:80) and so does the oracle.
    SURVEY.md measured RMS ~1e-8 between the reference's Release and strict builds; the oracle's two
    builds must sit in the same band, far below the 1e-4 parity bar."""
    strict = os.path.join(ROOT, "oracle", "liboracle_strict.so")
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "liboracle_strict.so"], check=True)
    for scene, tol in (("meshes.hexray", 1e-6), ("heightfield.hexray", 1e-5)):
        hs = hx.HostScene(scene_path(scene), hx.overlay_text(240, 135))
        a = oracle_render(hs, hx.frame_params(mode=hx.MODE_ADAPTIVE_AA))[0]
        b = oracle_render(hs, hx.frame_params(mode=hx.MODE_ADAPTIVE_AA), lib=oracle_lib(strict))[0]
        assert np.all(rms(a, b) < tol), (scene, rms(a, b))


@pytest.mark.parametrize("scene", ["boxed.hexray", "cornell_box.hexray", "kdtree_test.hexray", "hw10/bokeh.hexray", "hw12/sphtri.hexray", "zaphod.hexray", "beer.hexray"])
def test_other_shipped_scenes_render(scene):
    """Element classes outside the five configs (Cube, CSG, RectLight in raytrace mode, glossy Reflection ...)."""
    hs = hx.HostScene(scene_path(scene), hx.overlay_text(48, 36, num_paths=2, num_samples=2))
    img, cnt = oracle_render(hs, hx.frame_params())
    assert np.isfinite(img).all() and img.max() > 0 and cnt["camera_samples"] >= 48 * 36


def test_foveated_frame_structure():
    """Foveated pass 1 as restated from src/main.cpp:284-311: blocks whose centre is outside the radius are flat (one colour = the
    mean of five rays), blocks inside are the plain frame's pixels, AA only happens inside the fovea, and a radius that covers the
    frame changes nothing."""
    hs = hx.HostScene(scene_path("meshes.hexray"), hx.overlay_text(200, 150))
    plain, c0 = oracle_render(hs, hx.frame_params(mode=hx.MODE_SINGLE))
    vip, radius = (120, 70), 45
    fov, c1 = oracle_render(hs, hx.frame_params(mode=hx.MODE_SINGLE, foveated_radius=radius, vip=vip))
    coarse_blocks = fine_blocks = 0
    for by in range(0, 150, 8):
        for bx in range(0, 200, 8):
            blk, ref = fov[by:by + 8, bx:bx + 8], plain[by:by + 8, bx:bx + 8]
            if (bx + 4 - vip[0]) ** 2 + (by + 4 - vip[1]) ** 2 > radius ** 2:
                assert np.all(blk == blk[0, 0])
                coarse_blocks += 1
            else:
                assert np.array_equal(blk, ref)
                fine_blocks += 1
    assert coarse_blocks > fine_blocks > 10
    assert c1["camera_samples"] == 5 * coarse_blocks + sum(
        min(8, 200 - bx) * min(8, 150 - by) for by in range(0, 150, 8) for bx in range(0, 200, 8)
        if (bx + 4 - vip[0]) ** 2 + (by + 4 - vip[1]) ** 2 <= radius ** 2)
    aa, c2 = oracle_render(hs, hx.frame_params(mode=hx.MODE_ADAPTIVE_AA, foveated_radius=radius, vip=vip))
    ys, xs = np.nonzero(np.any(aa != fov, axis=2))
    assert len(ys) > 0 and np.all((xs - vip[0]) ** 2 + (ys - vip[1]) ** 2 <= radius ** 2)      # only pixels inside the fovea were re-sampled
    everything, _ = oracle_render(hs, hx.frame_params(mode=hx.MODE_SINGLE, foveated_radius=1000, vip=(100, 75)))
    assert np.array_equal(everything, plain)
    nothing_seen_yet, c3 = oracle_render(hs, hx.frame_params(mode=hx.MODE_SINGLE, foveated_radius=30))   # vip = (-100, -100): all coarse
    assert c3["camera_samples"] == 5 * 25 * 19


def test_prepass_blocks():
    """coarseRender as restated from src/main.cpp:387-414: 24x24 blocks, each the mean of `spp` rays; more rays, less noise."""
    hs = hx.HostScene(scene_path("meshes.hexray"), hx.overlay_text(120, 100))
    img, c = oracle_render(hs, hx.frame_params(mode=hx.MODE_PREPASS, spp=5))
    assert c["camera_samples"] == 5 * 5 * 5                      # 5 x 5 blocks (the last row and column are ragged), 5 rays each
    for by in range(0, 100, 24):
        for bx in range(0, 120, 24):
            blk = img[by:by + 24, bx:bx + 24]
            assert np.all(blk == blk[0, 0])
    full, _ = oracle_render(hs, hx.frame_params(mode=hx.MODE_SINGLE))
    means = np.stack([[full[by:by + 24, bx:bx + 24].mean(axis=(0, 1)) for bx in range(0, 120, 24)] for by in range(0, 100, 24)])
    e5 = np.abs(img[::24, ::24] - means).mean()
    many, _ = oracle_render(hs, hx.frame_params(mode=hx.MODE_PREPASS, spp=200))
    e200 = np.abs(many[::24, ::24] - means).mean()
    assert e200 < e5                                             # converges towards the block means of the real frame
