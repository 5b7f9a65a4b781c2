"""Developer aid: where does an adaptive-AA frame's time go?  Per-pass HIP-event times of meshes_aa in steady state."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import hexray_amd as hx
hx.init(0)
hs = hx.HostScene(os.path.join(ROOT, "data", "meshes.hexray"), hx.overlay_text(1920, 1080))
ds = hx.DeviceScene(hs)
p = hx.frame_params(tile_size=32)
for i in range(6):
    ds.render(p)
    t = ds.frame_timing()
    print("frame %d: total %.3f ms = render kernels %.3f + continuation %.3f + detect/resolve/gaps %.3f" % (i, t["total_ms"], t["render_ms"], t["continue_ms"], t["resolve_ms"]), flush=True)
one = hx.DeviceScene(hx.HostScene(os.path.join(ROOT, "data", "meshes.hexray"), hx.overlay_text(1920, 1080, want_aa=False)))
for i in range(4):
    one.render(p)
print("pass 1 alone (1 spp frame), steady: %.3f ms" % one.last_kernel_ms())
c = None
ds.render(hx.frame_params(tile_size=32, collect_counters=True))
print("flagged pixels:", ds.counters()["aa_pixels"])
