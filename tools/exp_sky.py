import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import hexray_amd as hx
hx.init(0)
for name, extra in (("normal", ""), ("sky", "Camera {\n pitch 80\n}\n"), ("floor", "Camera {\n pitch -80\n pos (60, 30, 60)\n}\n"), ("teapot-close", "Camera {\n pos (104, 30, 80)\n yaw 0\n pitch 0\n}\n")):
    hs = hx.HostScene(os.path.join(ROOT, "data", "meshes.hexray"), hx.overlay_text(1920, 1080, dof=True, num_samples=16) + extra)
    ds = hx.DeviceScene(hs)
    p = hx.frame_params(tile_size=32)
    ms = []
    for i in range(3):
        ds.render(p); ms.append(ds.last_kernel_ms())
    pc = hx.frame_params(tile_size=32, collect_counters=True)
    ds.render(pc); c = ds.counters()
    print(name, "kernel ms %.3f" % min(ms), "raycasts/sample %.2f shadow/sample %.2f tri/sample %.1f bbox/sample %.1f" % (c["raycasts"]/c["camera_samples"], c["shadow_queries"]/c["camera_samples"], c["tri_tests"]/c["camera_samples"], c["bbox_tests"]/c["camera_samples"]))
    ds.close()
