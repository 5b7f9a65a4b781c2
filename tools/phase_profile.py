"""Developer aid: per-phase wave time of the render kernel, from a -DHXR_PHASE_TIMERS build (tools/phase_profile.sh)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import hexray_amd as hx
import importlib.util
spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py")); bench = importlib.util.module_from_spec(spec); spec.loader.exec_module(bench)
names = os.environ.get("HXR_PHASE_NAMES", "dispense,gen(+reduce),query(nodes),mesh walk,leaf,shade").split(",")
hx.init(0)
for wl in sys.argv[1:] or ["meshes16"]:
    scene_name, ov, desc = bench.WORKLOADS[wl]
    hs = hx.HostScene(os.path.join(ROOT, "data", scene_name), hx.overlay_text(**ov))
    ds = hx.DeviceScene(hs, accel=hx.ACCEL_SAH if os.environ.get('HXR_ACCEL') == 'sah' else hx.ACCEL_REFERENCE)
    p = hx.frame_params(tile_size=32)
    if os.environ.get("HXR_SHARD"):   # "rank,count": profile one shard only (e.g. a single tile: count = number of tiles)
        import torch
        r, c = (int(v) for v in os.environ["HXR_SHARD"].split(","))
        p = hx.frame_params(tile_size=32, shard_rank=r, shard_count=c, layout=hx.LAYOUT_TILES)
        buf = torch.zeros(ds.output_bytes(p, 0) // 4 + 16, dtype=torch.float32, device="cuda")
    for i in range(3):
        if os.environ.get("HXR_SHARD"):
            ds.render_device(p, buf.data_ptr(), 0, torch.cuda.current_stream().cuda_stream)
            torch.cuda.synchronize()
        else:
            ds.render(p)
    ms = ds.last_kernel_ms()
    c = ds.counters_raw() if hasattr(ds, "counters_raw") else None
    ph = list(c.reserved_)[:6]
    tot = sum(ph) or 1
    print("%s: kernel %.3f ms; wave-clock shares:" % (wl, ms), "  ".join("%s %.1f%%" % (n, 100.0 * v / tot) for n, v in zip(names, ph)), " total clocks %.3e" % tot, " raw", ph, flush=True)
