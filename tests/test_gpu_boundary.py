"""GPU tests of the drop-in boundary beyond the frame itself (include/hexray_hip.h): the cancel flag and the progress callback of
hxr_render (reference render(): checkForUserExit per bucket row, displayVFBRect per finished bucket), the one-process multi-GPU
entry (hxr_group_*), and the `hexray` command-line host -- the only C++ caller of the ABI -- run as a child process."""
import ctypes
import os
import subprocess
import threading
import time

import numpy as np
import pytest

import hexray_amd as hx
from conftest import ROOT, scene_path

pytestmark = pytest.mark.gpu


def test_cancel_flag_stops_a_running_frame(gpu):
    """A frame that takes ~0.7 s (smallpt 1080p x 256 paths): the flag is raised 50 ms in, the call must come back long before the
    frame would have finished, with HXR_CANCELLED, and the scene handle must still render a complete frame afterwards."""
    hs = hx.HostScene(scene_path("smallpt.hexray"), hx.overlay_text(1920, 1080, num_paths=256))
    ds = hx.DeviceScene(hs)
    t0 = time.perf_counter()
    ds.render(hx.frame_params())
    full = time.perf_counter() - t0
    flag = ctypes.c_int(0)
    timer = threading.Timer(0.05, lambda: setattr(flag, "value", 1))
    timer.start()
    t0 = time.perf_counter()
    with pytest.raises(hx.Cancelled):
        ds.render(hx.frame_params(), cancel=flag)
    stopped = time.perf_counter() - t0
    timer.join()
    assert stopped < 0.6 * full, (stopped, full)           # the waves stopped taking packets: not the whole frame's work was done
    assert stopped < 0.05 + 0.15, stopped                  # one packet of 64 samples x 4 rounds per wave after the flag, not more
    raised_before = ctypes.c_int(1)
    with pytest.raises(hx.Cancelled):
        ds.render(hx.frame_params(), cancel=raised_before)
    small = hx.HostScene(scene_path("smallpt.hexray"), hx.overlay_text(96, 72, num_paths=8))
    a = hx.DeviceScene(small).render(hx.frame_params(), cancel=ctypes.c_int(0))   # a watched frame that is never cancelled is complete
    b = hx.DeviceScene(small).render(hx.frame_params())
    assert np.array_equal(a, b)
    ds.close()


def test_progress_callback_hands_over_finished_tiles(gpu):
    """Tiles are reported exactly once, cover the frame, and every reported tile already holds its final pixels in the caller's
    buffer at the time of the call -- while other tiles are still being rendered (Monte-Carlo frame); adaptive-AA frames report
    when the frame is final."""
    hs = hx.HostScene(scene_path("meshes.hexray"), hx.overlay_text(1920, 1080, dof=True, num_samples=16))
    ds = hx.DeviceScene(hs)
    final = ds.render(hx.frame_params(tile_size=64))
    out = np.full((1080, 1920, 3), -1.0, dtype=np.float32)
    seen, incomplete_elsewhere = [], []

    def on_tile(x0, y0, x1, y1):
        seen.append((x0, y0, x1, y1))
        assert np.array_equal(out[y0:y1, x0:x1], final[y0:y1, x0:x1])
        incomplete_elsewhere.append(bool((out == -1.0).any()))

    got = ds.render(hx.frame_params(tile_size=64), on_tile=on_tile, out=out)
    assert np.array_equal(got, final)
    cover = np.zeros((1080, 1920), dtype=np.int32)
    for (x0, y0, x1, y1) in seen:
        cover[y0:y1, x0:x1] += 1
    assert len(seen) == 30 * 17 and np.all(cover == 1)
    assert sum(incomplete_elsewhere) > len(seen) // 2                # most tiles arrived while the frame was still incomplete
    # adaptive AA: pixels are final only after pass 3, the callback still reports every tile once
    hs2 = hx.HostScene(scene_path("meshes.hexray"), hx.overlay_text(320, 200))
    d2 = hx.DeviceScene(hs2)
    rects = []
    f = d2.render(hx.frame_params(mode=hx.MODE_ADAPTIVE_AA, tile_size=32), on_tile=lambda *r: rects.append(r))
    assert len(rects) == 10 * 7 and np.array_equal(f, d2.render(hx.frame_params(mode=hx.MODE_ADAPTIVE_AA, tile_size=32)))
    ds.close()


def test_watched_frames_back_to_back_after_a_camera_move(gpu):
    """Two watched frames in a row with different cameras: the per-tile progress words of the first frame (every tile complete) must not
    be what the second frame's poll reads -- the clear is ordered before the first look (hxr_api.hip, "Watched frame") -- so every
    tile handed over during the second frame already holds the SECOND camera's pixels."""
    hs = hx.HostScene(scene_path("meshes.hexray"), hx.overlay_text(640, 360, dof=True, num_samples=16))
    ds = hx.DeviceScene(hs)
    p = hx.frame_params(tile_size=32)
    for step in range(3):
        if step:
            hs.move_camera(1.5 * step, 0.7)
            hs.rotate_camera(-6.0 * step, 2.0)
            ds.update_camera()
        final = ds.render(p)
        out = np.full((360, 640, 3), -1.0, dtype=np.float32)
        bad = []

        def on_tile(x0, y0, x1, y1):
            if not np.array_equal(out[y0:y1, x0:x1], final[y0:y1, x0:x1]):
                bad.append((x0, y0))

        got = ds.render(p, on_tile=on_tile, out=out)
        assert not bad, (step, bad[:4])
        assert np.array_equal(got, final)
    ds.close()


@pytest.mark.parametrize("devices", [[0, 0], [0, 0, 0]])
def test_device_group_equals_single_device(gpu, devices):
    """hxr_group_*: the C-level one-process multi-GPU driver.  This box has one GPU, so the group lists it several times (every
    entry gets its own scene replica, stream, shard and peer copy -- same code path, same-device copies): the assembled frame is
    the single-device frame bit for bit in Monte-Carlo, 1-spp and adaptive-AA modes, and cancel reaches every replica."""
    for ov, fp in ((hx.overlay_text(400, 300, dof=True, num_samples=16), dict()), (hx.overlay_text(333, 211, want_aa=False), dict(mode=hx.MODE_SINGLE)),
                   (hx.overlay_text(320, 240), dict(mode=hx.MODE_ADAPTIVE_AA))):
        hs = hx.HostScene(scene_path("meshes.hexray"), ov)
        whole = hx.DeviceScene(hs).render(hx.frame_params(**fp))
        grp = hx.DeviceGroup(hs, devices)
        tiles = []
        frame = grp.render(hx.frame_params(**fp), on_tile=lambda *r: tiles.append(r))
        assert np.array_equal(frame, whole)
        total, each = grp.last_ms()
        assert total > 0 and len(each) == len(devices) and all(e > 0 for e in each)
        assert len(tiles) == ((whole.shape[1] + 7) // 8) * ((whole.shape[0] + 7) // 8)
        with pytest.raises(hx.Cancelled):
            grp.render(hx.frame_params(**fp), cancel=ctypes.c_int(1))
        # hxr_group_create leaves the process-wide device selection as it found it: a plain scene created afterwards renders the same frame
        assert np.array_equal(hx.DeviceScene(hs).render(hx.frame_params(**fp)), whole)
        grp.close()


def test_hexray_binary_as_a_child_process(gpu, tmp_path):
    """The `hexray` executable (hexray_amd/host/main.cpp) in a fresh process: its raw frame equals the ctypes frame bit for bit,
    single device and `--devices 0,0` (the group driver), and it prints the reference's k-d tree statistics."""
    exe = os.path.join(ROOT, "hexray_amd", "bin", "hexray")
    assert os.path.exists(exe), "build.sh cli has not been run"
    hs = hx.HostScene(scene_path("meshes.hexray"), hx.overlay_text(320, 240, dof=True, num_samples=8))
    want = hx.DeviceScene(hs).render(hx.frame_params())
    for extra in ([], ["--devices", "0,0"], ["--progress", "--tile", "32"], ["--accel", "device-build"]):   # the last: kd-trees built on the GPU
        raw = tmp_path / "frame.raw"
        r = subprocess.run([exe, scene_path("meshes.hexray"), "--width", "320", "--height", "240", "--dof", "on", "--spp", "8", "--out", str(raw)] + extra,
                           capture_output=True, text=True, timeout=120, cwd=ROOT)
        assert r.returncode == 0, r.stdout + r.stderr
        assert "Exited cleanly" in r.stdout and "nodes            : 6007" in r.stdout
        if "device-build" in extra:
            assert "built on the GPU" in r.stdout and "leaf entries     : 33644" in r.stdout
        blob = np.fromfile(raw, dtype=np.uint8)
        w, h = np.frombuffer(blob[:8].tobytes(), dtype=np.int32)
        frame = np.frombuffer(blob[8:].tobytes(), dtype=np.float32).reshape(h, w, 3)
        assert (w, h) == (320, 240) and np.array_equal(frame, want), extra
        if "--progress" in extra:
            assert "tiles reported: 80" in r.stdout
        raw.unlink()
    # heightfield.hexray with every table of the heightfield prepared by the module: close to the host-prepared frame (float rounding of the blur)
    frames = []
    for extra in ([], ["--gpu-heightfield-blur"]):
        raw = tmp_path / "hf.raw"
        r = subprocess.run([exe, scene_path("heightfield.hexray"), "--width", "240", "--height", "180", "--aa", "off", "--out", str(raw)] + extra,
                           capture_output=True, text=True, timeout=120, cwd=ROOT)
        assert r.returncode == 0 and "Exited cleanly" in r.stdout, r.stdout + r.stderr
        frames.append(np.frombuffer(np.fromfile(raw, dtype=np.uint8)[8:].tobytes(), dtype=np.float32).reshape(180, 240, 3).astype(np.float64))
        raw.unlink()
    assert np.sqrt(np.mean((frames[0] - frames[1]) ** 2, axis=(0, 1))).max() <= 1e-4


def test_scene_tables_prepared_on_the_gpu(gpu):
    """hxr_scene_create builds the leaf-ordered hot triangle records on the GPU (both trees), and -- for a descriptor loaded without
    them -- the heightfield's maxH / highMap tables: pure copies and maxima, so every value must equal the host's (0 mismatches), and the
    frames must not change."""
    hs = hx.HostScene(scene_path("meshes.hexray"), hx.overlay_text(160, 120))
    for accel in (hx.ACCEL_REFERENCE, hx.ACCEL_SAH):
        ds = hx.DeviceScene(hs, accel=accel)
        assert ds.verify_tables() == 0
        ds.close()
    full = hx.HostScene(scene_path("heightfield.hexray"), hx.overlay_text(240, 135))
    lean = hx.HostScene(scene_path("heightfield.hexray"), hx.overlay_text(240, 135), gpu_heightfield_tables=True)
    from desc_types import desc_of
    assert desc_of(lean).heightfields[0].first_maxh == -1 and desc_of(full).heightfields[0].first_maxh >= 0
    assert desc_of(lean).f32_pool_size < desc_of(full).f32_pool_size - 512 * 512 * 16
    a, b = hx.DeviceScene(full), hx.DeviceScene(lean)
    assert b.verify_tables(full) == 0 and a.verify_tables() == 0
    p = hx.frame_params(mode=hx.MODE_ADAPTIVE_AA)
    assert np.array_equal(a.render(p), b.render(p))
    a.close(); b.close()


def test_malformed_mesh_tables_are_refused_before_any_kernel_sees_them(gpu):
    """A descriptor whose mesh indices point outside their tables must come back as HXR_E_INVALID from hxr_scene_create: the kernels
    follow these indices without checks (a stray one is a GPU fault, not an exception)."""
    hx = gpu
    from desc_types import desc_of
    hs = hx.HostScene(scene_path("meshes.hexray"), hx.overlay_text(width=64, height=48))
    d = desc_of(hs)
    lib = hx.hip_lib()

    def refused(what):
        s = lib.hxr_scene_create(hs.desc)
        if s:
            lib.hxr_scene_destroy(s)
        assert not s, what
        assert what in lib.hxr_last_error().decode(), (what, lib.hxr_last_error())

    m = d.meshes[1]
    node = d.kd_nodes[m.first_kd_node]                     # the root: a binary node
    keep = node.first
    node.first = 0                                         # a child that is its own ancestor
    refused("kd node 0 is malformed")
    node.first = int(m.num_kd_nodes)                       # children past the end
    refused("kd node 0 is malformed")
    node.first = keep
    keep = d.leaf_indices[m.first_leaf_index]
    d.leaf_indices[m.first_leaf_index] = int(m.num_triangles)
    refused("leaf entry 0 names triangle")
    d.leaf_indices[m.first_leaf_index] = keep
    tri = d.triangles[m.first_triangle + 5]
    keep = tri.v[1]
    tri.v[1] = int(m.num_vertices)
    refused("triangle 5 refers outside")
    tri.v[1] = keep
    keep = m.num_leaf_indices
    m.num_leaf_indices = int(d.num_leaf_indices) + 1
    refused("a table range lies outside the descriptor")
    m.num_leaf_indices = keep
    s = lib.hxr_scene_create(hs.desc)                      # restored: accepted again
    assert s
    lib.hxr_scene_destroy(s)


def test_malformed_heightfield_and_bitmap_ranges_are_refused(gpu):
    hx = gpu
    from desc_types import desc_of
    hs = hx.HostScene(scene_path("heightfield.hexray"), hx.overlay_text(width=64, height=48))
    d = desc_of(hs)
    lib = hx.hip_lib()

    def refused(what):
        s = lib.hxr_scene_create(hs.desc)
        if s:
            lib.hxr_scene_destroy(s)
        assert not s, what
        assert what in lib.hxr_last_error().decode(), (what, lib.hxr_last_error())

    h = d.heightfields[0]
    keep = h.first_highmap
    h.first_highmap = int(d.f32_pool_size) - 100           # the 16-level pyramid would run off the pool
    refused("heightfield 0: a table range lies outside")
    h.first_highmap = keep
    keep = h.max_k
    h.max_k = 17
    refused("heightfield 0: a table range lies outside")
    h.max_k = keep
    b = d.bitmaps[0]
    keep = b.first_texel
    b.first_texel = int(d.f32_pool_size) - 1
    refused("bitmap 0: texels outside")
    b.first_texel = keep
    keep = d.environment.side_bitmap[3]
    d.environment.side_bitmap[3] = int(d.num_bitmaps)
    refused("environment side 3")
    d.environment.side_bitmap[3] = keep
    s = lib.hxr_scene_create(hs.desc)
    assert s
    lib.hxr_scene_destroy(s)
