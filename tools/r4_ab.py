"""Round 4 A/B of a kernel build on the three figures the verdict asks for: steady-state frame, first frame of a fresh handle, slowest rank of an 8-way shard
(all meshes16, 1920x1080).  usage: HXR_HIP_LIB=<lib> python3 tools/r4_ab.py [workload]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import hexray_amd as hx
from hexray_amd.distributed import padded_tiles_per_rank
import importlib.util
spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py")); bench = importlib.util.module_from_spec(spec); spec.loader.exec_module(bench)
wl = sys.argv[1] if len(sys.argv) > 1 else "meshes16"
scene_name, ov, _ = bench.WORKLOADS[wl]
hx.init(0)
hs = hx.HostScene(os.path.join(ROOT, "data", scene_name), hx.overlay_text(**ov))
W, H = hs.settings.frame_width, hs.settings.frame_height
st = torch.cuda.current_stream().cuda_stream
frame = torch.zeros((H, W, 3), dtype=torch.float32, device="cuda")
p = hx.frame_params(tile_size=64)
accel = hx.ACCEL_SAH if os.environ.get("HXR_ACCEL") == "sah" else hx.ACCEL_REFERENCE
ds = hx.DeviceScene(hs, accel=accel)
ms = []
for i in range(13):
    ds.render_device(p, frame.data_ptr(), W * 12, st); torch.cuda.synchronize()
    if i >= 3: ms.append(ds.last_kernel_ms())
steady = float(np.mean(ms))
first = []
for _ in range(5):
    c = hx.DeviceScene(hs, accel=accel)
    c.render_device(p, frame.data_ptr(), W * 12, st); torch.cuda.synchronize()
    first.append(c.last_kernel_ms()); c.close()
out = "%-28s steady %.3f (min %.3f max %.3f)  first %.3f (min %.3f)" % (os.path.basename(os.environ.get("HXR_HIP_LIB", "libhexray_hip.so")), steady, min(ms), max(ms), float(np.median(first)), min(first))
if not os.environ.get("HXR_AB_NO_SHARDS"):
    T, world = 8, 8
    slots = padded_tiles_per_rank(W, H, T, world)
    buf = torch.zeros(slots * T * T * 3, dtype=torch.float32, device="cuda")
    ranks = []
    for rank in range(world):
        q = hx.frame_params(tile_size=T, shard_rank=rank, shard_count=world, layout=hx.LAYOUT_TILES)
        r = []
        for i in range(5):
            ds.render_device(q, buf.data_ptr(), 0, st); torch.cuda.synchronize()
            if i >= 2: r.append(ds.last_kernel_ms())
        ranks.append(min(r))
    out += "  8-way ranks: slowest %.3f sum %.2f" % (max(ranks), sum(ranks))
print(out, flush=True)
