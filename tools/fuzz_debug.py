"""Developer aid: which generated feature makes a fuzz scene (tests/test_gpu_fuzz.py) differ between GPU and oracle."""
import os, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import hexray_amd as hx
from oracle_binding import oracle_render
import importlib.util
spec = importlib.util.spec_from_file_location("t", os.path.join(ROOT, "tests", "test_gpu_fuzz.py")); t = importlib.util.module_from_spec(spec); spec.loader.exec_module(t)
hx.init(0)
d = tempfile.mkdtemp()
for seed in [int(a) for a in sys.argv[1:]]:
    for without in ((), ("bump",), ("wood",), ("env",), ("hf",), ("bitmaps",)):
        text, mode = t.make_scene(seed, os.path.relpath(os.path.join(ROOT, "data", "geom"), d), without)
        p = os.path.join(d, "s.hexray"); open(p, "w").write(text)
        hs = hx.HostScene(p)
        fp = hx.frame_params(collect_counters=True)
        ds = hx.DeviceScene(hs)
        got = ds.render(fp); gc = ds.counters()
        want, oc = oracle_render(hs, fp)
        rms = np.sqrt(np.mean((got.astype(np.float64) - want) ** 2, axis=(0, 1))).max()
        bad = {k: (gc[k], oc[k]) for k in oc if k in gc and gc[k] != oc[k] and k not in ("mesh_hits", "hf_highest", "hf_height", "hf_normals", "leaf_indices")}
        print(seed, mode, "without", without, "rms %.2e" % rms, "counters", bad, flush=True)
