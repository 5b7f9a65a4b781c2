"""Developer aid (-DHXR_TURN_STATS build): how full are the turns of the path tracer's state machine?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import hexray_amd as hx
import importlib.util
spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py")); bench = importlib.util.module_from_spec(spec); spec.loader.exec_module(bench)
hx.init(0)
for wl in sys.argv[1:] or ["smallpt16", "cornell16"]:
    scene_name, ov, desc = bench.WORKLOADS[wl]
    hs = hx.HostScene(os.path.join(ROOT, "data", scene_name), hx.overlay_text(**ov))
    ds = hx.DeviceScene(hs, adaptive_schedule=False)
    ds.render(hx.frame_params(tile_size=32))
    r = list(ds.counters_raw().reserved_)
    turns, alive, served, cturns, cserved = r[0], r[1], r[2], r[3], r[4]
    print("%s: %d turns; lanes still tracing per turn %.1f, lanes served per turn %.1f; closest turns %d with %.1f lanes, shadow turns %d with %.1f lanes" %
          (wl, turns, alive / max(1, turns), served / max(1, turns), cturns, cserved / max(1, cturns), turns - cturns, (served - cserved) / max(1, turns - cturns)), flush=True)
