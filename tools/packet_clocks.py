"""Developer aid (-DHXR_PHASE_TIMERS -DHXR_PACKET_CLOCKS build): distribution of packet durations, whole frame and one shard."""
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
accel = hx.ACCEL_SAH if os.environ.get('HXR_ACCEL') == 'sah' else hx.ACCEL_REFERENCE
ds = hx.DeviceScene(hs, accel=accel)
print('accel', accel)
W, H, T = hs.settings.frame_width, hs.settings.frame_height, 32
for world, rank in ((1, 0), (8, 0), (8, 4)):
    slots = padded_tiles_per_rank(W, H, T, world)
    buf = torch.zeros(slots * T * T * 3, dtype=torch.float32, device="cuda")
    p = hx.frame_params(tile_size=T, shard_rank=rank, shard_count=world, layout=hx.LAYOUT_TILES)
    for i in range(3):
        ds.render_device(p, buf.data_ptr(), 0, torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
    ms = ds.last_kernel_ms()
    r = list(ds.counters_raw().reserved_)
    print("world %d rank %d: kernel %.3f ms; packets %d, mean %.0f clocks, max %d clocks; > 250k: %d, > 1M: %d, > 4M: %d" % (world, rank, ms, r[2], r[0] / max(1, r[2]), r[1], r[5], r[3], r[4]), flush=True)
