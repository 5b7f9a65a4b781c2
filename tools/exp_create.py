"""Developer aid: how long do scene preparation and hxr_scene_create take (first and later calls)?"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import hexray_amd as hx
hx.init(0)
for name in (sys.argv[1:] or ["meshes.hexray", "heightfield.hexray", "simple.hexray"]):
    for rep in range(3):
        t0 = time.perf_counter()
        hs = hx.HostScene(os.path.join(ROOT, "data", name), hx.overlay_text(1920, 1080))
        t1 = time.perf_counter()
        ds = hx.DeviceScene(hs)
        t2 = time.perf_counter()
        ds2 = hx.DeviceScene(hs, accel=hx.ACCEL_SAH)
        t3 = time.perf_counter()
        print(name, "host %.1f ms, hxr_scene_create %.1f ms, with SAH rebuild %.1f ms" % ((t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3), flush=True)
        ds.close(); ds2.close()

# ... and with the kd-trees (and a heightfield's blur / normals / pyramid) left to the module
for name in ("meshes.hexray", "heightfield.hexray"):
    for rep in range(3):
        t0 = time.perf_counter()
        bare = hx.HostScene(os.path.join(ROOT, "data", name), gpu_kd_build=True, gpu_heightfield_blur=True)
        t1 = time.perf_counter()
        ds = hx.DeviceScene(bare, accel=hx.ACCEL_REFERENCE_DEVICE_BUILD)
        t2 = time.perf_counter()
        print("%s prepared on the GPU: host %.1f ms, hxr_scene_create %.1f ms" % (name, (t1 - t0) * 1e3, (t2 - t1) * 1e3), flush=True)
        ds.close()
