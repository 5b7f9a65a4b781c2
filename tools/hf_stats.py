"""Developer aid (-DHXR_HF_STATS build): lane utilisation inside the heightfield marcher (lane-steps taken / lane-steps offered)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import hexray_amd as hx
import importlib.util
spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py")); bench = importlib.util.module_from_spec(spec); spec.loader.exec_module(bench)
hx.init(0)
for wl in sys.argv[1:] or ["heightfield16"]:
    scene_name, ov, desc = bench.WORKLOADS[wl]
    hs = hx.HostScene(os.path.join(ROOT, "data", scene_name), hx.overlay_text(**ov))
    ds = hx.DeviceScene(hs)
    for i in range(2):
        ds.render(hx.frame_params(tile_size=32, collect_counters=True))
    c = ds.counters()
    print("%s: hf calls %d, steps offered per call %.1f, taken %.1f (%.0f%% of the lane-steps do work); pyramid probes/call %.1f, triangle pairs/call %.2f" %
          (wl, c["hf_calls"], c["kd_visits"] / c["hf_calls"], c["box_test_mismatch"] / c["hf_calls"], 100.0 * c["box_test_mismatch"] / max(1, c["kd_visits"]),
           c["hf_highest"] / c["hf_calls"], c["hf_tri_tests"] / 2 / c["hf_calls"]), flush=True)
