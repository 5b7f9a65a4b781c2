"""Developer aid: build time of the module's SAH trees on the device against the host builder, per mesh, and hxr_scene_create_ex end to end."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import hexray_amd as hx
from desc_types import SceneDesc
hx.init(0)
for scene in ("data/meshes.hexray", "tests/golden/all_meshes.hexray"):
    hs = hx.HostScene(os.path.join(ROOT, scene))
    d = SceneDesc.from_address(hs.desc)
    hx.sah_build(hs, 0, device=True)   # first call: code objects
    for mi in range(d.num_meshes):
        best = min(hx.sah_build(hs, mi, device=True)[3] for _ in range(3))
        t = time.perf_counter(); nodes, leaf, depth, _ = hx.sah_build(hs, mi, device=False); host_ms = (time.perf_counter() - t) * 1e3
        print("%s mesh %d: %d triangles, %d nodes, %d leaf entries, depth %d: device %.2f ms, host %.1f ms" % (scene, mi, d.meshes[mi].num_triangles, len(nodes), len(leaf), depth, best, host_ms), flush=True)
    for accel, name in ((hx.ACCEL_REFERENCE, "reference trees as handed over"), (hx.ACCEL_SAH, "SAH trees built on the device")):
        hx.DeviceScene(hs, accel=accel).close()
        ts = []
        for _ in range(3):
            t = time.perf_counter(); hx.DeviceScene(hs, accel=accel).close(); ts.append((time.perf_counter() - t) * 1e3)
        print("%s hxr_scene_create_ex + destroy, %s: %.2f ms" % (scene, name, min(ts)), flush=True)
    os.environ["HXR_SAH_HOST"] = "1"
    ts = []
    for _ in range(3):
        t = time.perf_counter(); hx.DeviceScene(hs, accel=hx.ACCEL_SAH).close(); ts.append((time.perf_counter() - t) * 1e3)
    print("%s hxr_scene_create_ex + destroy, SAH trees built on the host: %.2f ms" % (scene, min(ts)), flush=True)
    del os.environ["HXR_SAH_HOST"]
