"""Developer aid: how much of the headline frame is secondary rays?  meshes16 with maxTraceDepth 7 (as shipped) / 0
(reflection rays end black without being traced), adaptive schedule on / off, first frames and steady state."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import hexray_amd as hx
hx.init(0)
for depth in (7, 0):
    for adaptive in (True, False):
        ov = hx.overlay_text(width=1920, height=1080, dof=True, num_samples=16, extra="GlobalSettings {\nmaxTraceDepth %d\n}\n" % depth)
        hs = hx.HostScene(os.path.join(ROOT, "data", "meshes.hexray"), ov)
        ds = hx.DeviceScene(hs, adaptive_schedule=adaptive)
        p = hx.frame_params(tile_size=32)
        ms = []
        for i in range(6):
            ds.render(p)
            ms.append(ds.last_kernel_ms())
        print("maxTraceDepth", depth, "adaptive", adaptive, "kernel ms:", " ".join("%.3f" % m for m in ms), flush=True)
        ds.close()
