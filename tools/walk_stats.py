"""Developer aid (-DHXR_LEAF_STATS / -DHXR_LEAF_HIST / -DHXR_WALK_STATS builds, one at a time): the six developer counters of a frame, raw.
usage: HXR_HIP_LIB=hexray_amd/lib/libhexray_hip_<tag>.so python3 tools/walk_stats.py [workload ...]"""
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
    ds = hx.DeviceScene(hs, accel=hx.ACCEL_REFERENCE, adaptive_schedule=False)
    ds.render(hx.frame_params(tile_size=64))
    print(wl, os.environ.get("HXR_HIP_LIB", "default"), "dev counters:", list(ds.counters_raw().reserved_), flush=True)
