"""Developer aid: frame time of a workload against hxr_frame_params::tile_size (single GPU: the tile size only orders the packets)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import hexray_amd as hx
import importlib.util
spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py")); bench = importlib.util.module_from_spec(spec); spec.loader.exec_module(bench)
hx.init(0)
for wl in sys.argv[1:] or ["meshes16"]:
    scene_name, ov, desc = bench.WORKLOADS[wl]
    hs = hx.HostScene(os.path.join(ROOT, "data", scene_name), hx.overlay_text(**ov))
    ds = hx.DeviceScene(hs)
    for tile in (8, 16, 32, 64, 128, 0):
        ms = []
        for i in range(6):
            ds.render(hx.frame_params(tile_size=tile)); ms.append(ds.last_kernel_ms())
        print("%s tile %3d: %s" % (wl, tile, " ".join("%.3f" % m for m in ms[2:])), flush=True)
