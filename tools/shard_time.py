"""Developer aid: kernel time of one rank's share of the frame for world sizes 1, 2, 4, 8 (all on this GPU, one at a time)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import hexray_amd as hx
from hexray_amd.distributed import padded_tiles_per_rank
import importlib.util
spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py")); bench = importlib.util.module_from_spec(spec); spec.loader.exec_module(bench)
wl = sys.argv[1] if len(sys.argv) > 1 else "meshes16"
scene_name, ov, desc = bench.WORKLOADS[wl]
hs = hx.HostScene(os.path.join(ROOT, "data", scene_name), hx.overlay_text(**ov))
hx.init(0)
ds = hx.DeviceScene(hs, accel=hx.ACCEL_SAH if os.environ.get('HXR_ACCEL') == 'sah' else hx.ACCEL_REFERENCE)
W, H, T = hs.settings.frame_width, hs.settings.frame_height, int(os.environ.get("HXR_TILE", "32"))
for world in (1, 2, 4, 8):
    slots = padded_tiles_per_rank(W, H, T, world)
    buf = torch.zeros(slots * T * T * 3, dtype=torch.float32, device="cuda")
    for rank in (range(world) if os.environ.get("HXR_ALL_RANKS") else sorted({0, world // 2, world - 1})):
        p = hx.frame_params(tile_size=T, shard_rank=rank, shard_count=world, layout=hx.LAYOUT_TILES)
        ms = []
        for i in range(5):
            ds.render_device(p, buf.data_ptr(), 0, torch.cuda.current_stream().cuda_stream)
            torch.cuda.synchronize()
            ms.append(ds.last_kernel_ms())
        print("world %d rank %d: kernel ms %s  (x%d = %.2f)" % (world, rank, " ".join("%.3f" % m for m in ms[1:]), world, world * min(ms[1:])), flush=True)
