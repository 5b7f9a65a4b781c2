"""Developer aid: kernel time of the FIRST frame of fresh scene handles (no cost map) and of the steady state, per workload."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import hexray_amd as hx
import importlib.util
spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py")); bench = importlib.util.module_from_spec(spec); spec.loader.exec_module(bench)
hx.init(0)
for wl in (sys.argv[1:] or ["meshes16", "meshes1", "meshes_aa"]):
    scene_name, ov, desc = bench.WORKLOADS[wl]
    hs = hx.HostScene(os.path.join(ROOT, "data", scene_name), hx.overlay_text(**ov))
    warm = hx.DeviceScene(hs)
    p = hx.frame_params(tile_size=32)
    for _ in range(4):
        ref = warm.render(p)
    steady = warm.last_kernel_ms()
    first = []
    for _ in range(4):
        ds = hx.DeviceScene(hs)
        f = ds.render(p)
        first.append(ds.last_kernel_ms())
        assert np.array_equal(f, ref)
        ds.close()
    print("%s: first frame %s ms (median %.3f), steady %.3f ms, ratio %.2f" % (wl, " ".join("%.3f" % m for m in first), float(np.median(first)), steady, float(np.median(first)) / steady), flush=True)
