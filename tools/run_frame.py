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
if os.environ.get("HXR_OV"):   # e.g. HXR_OV="width=960,height=540,num_samples=64": override entries of the workload's overlay
    ov = dict(ov, **{k: int(v) for k, v in (kv.split("=") for kv in os.environ["HXR_OV"].split(","))})
hs = hx.HostScene(os.path.join(ROOT, "data", scene_name), hx.overlay_text(**ov, extra=os.environ.get("HXR_EXTRA_OVERLAY", "")))   # HXR_EXTRA_OVERLAY: more scene text for experiments
hx.init(0)
ds = hx.DeviceScene(hs, accel=hx.ACCEL_SAH if os.environ.get('HXR_ACCEL') == 'sah' else hx.ACCEL_REFERENCE)
p = hx.frame_params(tile_size=32)
shard = os.environ.get("HXR_SHARD")   # "rank,count": render one shard only (a single tile when count = number of tiles)
if shard:
    import torch
    r, c = (int(v) for v in shard.split(","))
    p = hx.frame_params(tile_size=32, shard_rank=r, shard_count=c, layout=hx.LAYOUT_TILES)
    buf = torch.zeros(ds.output_bytes(p, 0) // 4 + 16, dtype=torch.float32, device="cuda")
ms = []
for i in range(frames):
    if shard:
        ds.render_device(p, buf.data_ptr(), 0, torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
    else:
        ds.render(p)
    ms.append(ds.last_kernel_ms())
print(wl, "kernel ms:", " ".join("%.3f" % m for m in ms))
