"""Developer aid: frame timing of a bench workload with the deferred-ray pipeline on / off (hxr_get_frame_timing)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import hexray_amd as hx
import importlib.util
spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py")); bench = importlib.util.module_from_spec(spec); spec.loader.exec_module(bench)
hx.init(0)
frames_n = int(os.environ.get("FRAMES", "6"))
for wl in (sys.argv[1:] or ["meshes16"]):
    scene_name, ov, desc = bench.WORKLOADS[wl]
    hs = hx.HostScene(os.path.join(ROOT, "data", scene_name), hx.overlay_text(**ov))
    ref = None
    for deferred, adaptive in ((True, True), (False, True), (False, False)):
        ds = hx.DeviceScene(hs, deferred_rays=deferred, adaptive_schedule=adaptive)
        p = hx.frame_params(tile_size=32)
        rows = []
        for i in range(frames_n):
            f = ds.render(p)
            t = ds.frame_timing()
            rows.append(t)
        if ref is None:
            ref = f
        d = np.abs(f.astype(np.float64) - ref).max()
        print(wl, "deferred" if deferred else "in-place", "adaptive" if adaptive else "frame-order",
              "total ms:", " ".join("%.3f" % r["total_ms"] for r in rows),
              "| last: render %.3f continue %.3f resolve %.3f rays %d | max abs diff vs first config %.3g" % (rows[-1]["render_ms"], rows[-1]["continue_ms"], rows[-1]["resolve_ms"], rows[-1]["deferred_rays"], d), flush=True)
        ds.close()
