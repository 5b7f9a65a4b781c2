"""Developer aid: frame time and work counters of the reference kd-tree against the SAH rebuild (hxr_scene_create_ex)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import hexray_amd as hx
import importlib.util
spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py")); bench = importlib.util.module_from_spec(spec); spec.loader.exec_module(bench)
hx.init(0)
for wl in sys.argv[1:] or ["meshes16"]:
    scene_name, ov, desc = bench.WORKLOADS[wl]
    hs = hx.HostScene(os.path.join(ROOT, "data", scene_name), hx.overlay_text(**ov))
    frames = {}
    for name, accel in (("reference", hx.ACCEL_REFERENCE), ("sah", hx.ACCEL_SAH)):
        ds = hx.DeviceScene(hs, accel=accel)
        p = hx.frame_params(tile_size=32)
        ms = []
        for i in range(5):
            frames[name] = ds.render(p)
            ms.append(ds.last_kernel_ms())
        ds.render(hx.frame_params(tile_size=32, collect_counters=True))
        c = ds.counters()
        n = c["camera_samples"]
        stats = [ds.tree_stats(m) for m in range(8) if m < hs.num_meshes] if hasattr(hs, "num_meshes") else []
        print("%s %-9s kernel ms %s | per sample: bbox %.2f kd %.2f tri %.2f %s" % (wl, name, " ".join("%.3f" % m for m in ms[1:]), c["bbox_tests"] / n, c["kd_visits"] / n,
              c["tri_tests"] / n, stats), flush=True)
    d = frames["sah"].astype(np.float64) - frames["reference"]
    print("   sah vs reference tree: per-channel RMS", np.sqrt((d ** 2).mean(axis=(0, 1))), "pixels differing by > 1e-5:", int((np.abs(d).max(axis=2) > 1e-5).sum()), flush=True)
