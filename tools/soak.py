"""Developer aid: render the same frame many times (adaptive schedule on) and require every frame to be bit-identical."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import hexray_amd as hx
import importlib.util
spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py")); bench = importlib.util.module_from_spec(spec); spec.loader.exec_module(bench)
wl = sys.argv[1] if len(sys.argv) > 1 else "meshes16"
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 200
scene_name, ov, desc = bench.WORKLOADS[wl]
hs = hx.HostScene(os.path.join(ROOT, "data", scene_name), hx.overlay_text(**ov))
hx.init(0)
W, H = hs.settings.frame_width, hs.settings.frame_height
for world, rank, tile in ((1, 0, 32), (8, 0, 8)):
    ds = hx.DeviceScene(hs, accel=hx.ACCEL_SAH if os.environ.get("HXR_ACCEL") == "sah" else hx.ACCEL_REFERENCE)
    p = hx.frame_params(tile_size=tile, shard_rank=rank, shard_count=world, layout=hx.LAYOUT_TILES if world > 1 else hx.LAYOUT_FRAME)
    n = ds.output_bytes(p, W * 12) // 4
    a = torch.zeros(n, dtype=torch.float32, device="cuda")
    b = torch.zeros(n, dtype=torch.float32, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    ds.render_device(p, a.data_ptr(), W * 12, st)
    torch.cuda.synchronize()
    bad = 0
    for i in range(frames):
        b.zero_()
        ds.render_device(p, b.data_ptr(), W * 12, st)
        torch.cuda.synchronize()
        if not torch.equal(a, b):
            bad += 1
            print("frame", i, "differs in", int((a != b).sum()), "floats", flush=True)
    print("%s world %d: %d frames, %d differ from the first" % (wl, world, frames, bad), flush=True)
