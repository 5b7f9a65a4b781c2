"""Developer aid (-DHXR_PACKET_CLOCK_IMAGE build): per-pixel packet duration of a workload, saved under gpurun_out/."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import hexray_amd as hx
import importlib.util
spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py")); bench = importlib.util.module_from_spec(spec); spec.loader.exec_module(bench)
wl = sys.argv[1] if len(sys.argv) > 1 else "meshes16"
scene_name, ov, desc = bench.WORKLOADS[wl]
hs = hx.HostScene(os.path.join(ROOT, "data", scene_name), hx.overlay_text(**ov))
hx.init(0)
ds = hx.DeviceScene(hs)
f = ds.render(hx.frame_params(tile_size=32))[:, :, 0]
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
small = f.reshape(f.shape[0] // 2, 2, f.shape[1] // 2, 2).max(axis=(1, 3))   # one value per 2x2 quad = packet
np.save(os.path.join(ROOT, "gpurun_out", "cost_map_%s.npy" % wl), small.astype(np.float32))
print("packets", small.size, "mean", small.mean(), "max", small.max())
ys, xs = np.unravel_index(np.argsort(small.ravel())[::-1][:30], small.shape)
print("worst packets (x, y in pixels; clocks):", [(int(x) * 2, int(y) * 2, int(small[y, x])) for x, y in zip(xs, ys)])
