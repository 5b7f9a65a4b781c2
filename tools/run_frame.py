"""Developer aid for profiling: render N frames of a bench workload through the C ABI (no torch, no oracle)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import hexray_amd as hx
sys.argv += [None] * 3
wl = sys.argv[1] or "meshes16"
frames = int(sys.argv[2] or 3)
import importlib.util
spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py")); bench = importlib.util.module_from_spec(spec); spec.loader.exec_module(bench)
scene_name, ov, desc = bench.WORKLOADS[wl]
hs = hx.HostScene(os.path.join(ROOT, "data", scene_name), hx.overlay_text(**ov))
hx.init(0)
ds = hx.DeviceScene(hs)
p = hx.frame_params(tile_size=32)
ms = []
for i in range(frames):
    ds.render(p)
    ms.append(ds.last_kernel_ms())
print(wl, "kernel ms:", " ".join("%.3f" % m for m in ms))
