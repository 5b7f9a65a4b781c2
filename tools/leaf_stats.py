"""Developer aid (-DHXR_LEAF_STATS build): lanes taking part in leaf visits of the packet walk."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import hexray_amd as hx
import importlib.util
spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py")); bench = importlib.util.module_from_spec(spec); spec.loader.exec_module(bench)
hx.init(0)
for wl in sys.argv[1:] or ["meshes16"]:
    scene_name, ov, desc = bench.WORKLOADS[wl]
    hs = hx.HostScene(os.path.join(ROOT, "data", scene_name), hx.overlay_text(**ov))
    for accel in (hx.ACCEL_REFERENCE, hx.ACCEL_SAH):
        ds = hx.DeviceScene(hs, accel=accel, adaptive_schedule=False)
        ds.render(hx.frame_params(tile_size=32))
        r = list(ds.counters_raw().reserved_)
        print("%s accel %d: leaf visits %d, triangle iterations (wave level) %d = %.1f per visit, lanes per visit %.1f; walk steps %d of which empty-leaf visits %d (%.1f%%); "
              "cost if every visit took min(count, 2 + 1.5 lanes): %.1f%% of now" % (wl, accel, r[2], r[0], r[0] / max(1, r[2]), r[3] / max(1, r[2]), r[5], r[4], 100.0 * r[4] / max(1, r[5]), 100.0 * r[1] / max(1, r[0])), flush=True)
