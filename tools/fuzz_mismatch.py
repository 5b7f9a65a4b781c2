"""Developer aid: which generated scenes of tests/test_gpu_fuzz.py have work counters that differ from the oracle's, and by how much.
usage: python3 tools/fuzz_mismatch.py [first] [count]"""
import os, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import hexray_amd as hx
from oracle_binding import oracle_render
from test_gpu_fuzz import make_scene
first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
count = int(sys.argv[2]) if len(sys.argv) > 2 else 600
hx.init(0)
keys = ("raycasts", "shadow_queries", "node_tests", "bbox_tests", "kd_visits", "tri_tests", "tex_bilinear", "env_bilinear", "hf_calls", "hf_tri_tests")
with tempfile.TemporaryDirectory() as tmp:
    for seed in range(first, first + count):
        text, mode = make_scene(seed, os.path.relpath(os.path.join(ROOT, "data", "geom"), tmp))
        path = os.path.join(tmp, "fuzz_%d.hexray" % seed)
        open(path, "w").write(text)
        hs = hx.HostScene(path)
        p = hx.frame_params(collect_counters=True)
        ds = hx.DeviceScene(hs)
        got = ds.render(p)
        gc = ds.counters()
        want, oc = oracle_render(hs, p)
        rms = np.sqrt(np.mean((got.astype(np.float64) - want) ** 2, axis=(0, 1))).max()
        delta = {k: gc[k] - oc[k] for k in keys if gc[k] != oc[k]}
        if delta or rms > 1e-5:
            print("seed", seed, mode, "rms %.3g" % rms, "delta (gpu - oracle)", delta, "oracle", {k: oc[k] for k in delta}, flush=True)
        ds.close(); hs.close()
print("done", first, count)
