"""Round 4: what one rank of an N-way frame costs per frame when frames follow each other on the stream without a host synchronisation (the bench loop at N > 1,
minus the gather): the period between frames against the render kernel's own duration -- the difference is the schedule kernels, the clears and the launch gaps.
usage: python3 tools/exp_shard_period.py [world] [frames]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import hexray_amd as hx
from hexray_amd.distributed import padded_tiles_per_rank
hx.init(0)
hs = hx.HostScene(os.path.join(ROOT, "data", "meshes.hexray"), hx.overlay_text(width=1920, height=1080, dof=True, num_samples=16))
world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 200
T = 8 if world > 1 else 64
slots = padded_tiles_per_rank(1920, 1080, T, world)
buf = torch.zeros(slots * T * T * 3, dtype=torch.float32, device="cuda")
st = torch.cuda.current_stream().cuda_stream
ds = hx.DeviceScene(hs)
p = hx.frame_params(tile_size=T, shard_rank=0, shard_count=world, layout=hx.LAYOUT_TILES)
for _ in range(10): ds.render_device(p, buf.data_ptr(), 0, st)
torch.cuda.synchronize()
for rep in range(3):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    for _ in range(frames): ds.render_device(p, buf.data_ptr(), 0, st)
    e1.record()
    t_host = time.perf_counter() - t0
    torch.cuda.synchronize()
    print("world %d rank 0: period %.4f ms per frame on the stream (host enqueue %.4f ms per frame); render kernel alone %.4f ms" % (world, e0.elapsed_time(e1) / frames, t_host / frames * 1e3, ds.last_kernel_ms()), flush=True)
