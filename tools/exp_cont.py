"""Developer aid: continuation-pass timing of meshes16 at several maxTraceDepth values (chains vs bulk)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import hexray_amd as hx
hx.init(0)
for depth in (int(a) for a in (sys.argv[1:] or ["7", "1", "2"])):
    ov = hx.overlay_text(width=1920, height=1080, dof=True, num_samples=16, extra="GlobalSettings {\nmaxTraceDepth %d\n}\n" % depth)
    hs = hx.HostScene(os.path.join(ROOT, "data", "meshes.hexray"), ov)
    ds = hx.DeviceScene(hs)
    p = hx.frame_params(tile_size=32)
    for i in range(3):
        ds.render(p)
        t = ds.frame_timing()
    print("maxTraceDepth", depth, "total %.3f render %.3f continue %.3f rays %d" % (t["total_ms"], t["render_ms"], t["continue_ms"], t["deferred_rays"]), flush=True)
    ds.close()
