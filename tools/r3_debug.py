"""Developer aid: are frames reproducible from frame to frame (the adaptive schedule changes the packets' composition), and do they match the oracle?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import hexray_amd as hx
from oracle_binding import oracle_render
hx.init(0)
def diff(a, b):
    d = np.abs(a.astype(np.float64) - b).max(axis=2)
    ys, xs = np.nonzero(d > 0)
    return int((d > 0).sum()), float(d.max()), (list(zip(xs[:5].tolist(), ys[:5].tolist())))
for scene, ov in (("meshes.hexray", dict(width=640, height=360, dof=True, num_samples=16)), ("meshes.hexray", dict(width=320, height=200)), ("cornell_box.hexray", dict(width=320, height=180, num_paths=16))):
    hs = hx.HostScene(os.path.join(ROOT, "data", scene), hx.overlay_text(**ov))
    p = hx.frame_params(tile_size=32)
    a = hx.DeviceScene(hs)
    f = [a.render(p) for _ in range(3)]
    b = hx.DeviceScene(hs, adaptive_schedule=False)
    g = [b.render(p) for _ in range(2)]
    print(scene, ov, "sched f0-f1", diff(f[0], f[1]), "f1-f2", diff(f[1], f[2]), "nosched g0-g1", diff(g[0], g[1]), "f0-g0", diff(f[0], g[0]), flush=True)
    want, oc = oracle_render(hs, hx.frame_params(collect_counters=True))
    got = a.render(hx.frame_params(tile_size=32, collect_counters=True))
    gc = a.counters()
    rms = np.sqrt(np.mean((got.astype(np.float64) - want) ** 2, axis=(0, 1)))
    print("  vs oracle rms", rms, "diff counted-frame vs f0", diff(got, f[0]))
    for k in ("raycasts", "shadow_queries", "node_tests", "bbox_tests", "kd_visits", "tri_tests", "mesh_hits"):
        print("   ", k, gc[k], oc[k], "" if gc[k] == oc[k] else "<<<<<< MISMATCH")
