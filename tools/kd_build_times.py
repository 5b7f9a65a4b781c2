"""Developer aid: build time of the reference's midpoint kd-tree on the device (hxr_kd_build_device), per mesh of the scenes given."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import hexray_amd as hx
from desc_types import SceneDesc
hx.init(0)
scenes = sys.argv[1:] or ["data/meshes.hexray", "tests/golden/all_meshes.hexray"]
for scene in scenes:
    hs = hx.HostScene(scene if os.path.isabs(scene) else os.path.join(ROOT, scene))
    d = SceneDesc.from_address(hs.desc)
    hx.kd_build_device(hs, 0)   # first call: code objects
    for mi in range(d.num_meshes):
        runs = [hx.kd_build_device(hs, mi) for _ in range(5)]
        print("%s mesh %d: %d triangles, %d nodes, %d leaf entries, depth %d: device build %.2f ms (best of 5)" % (
            os.path.basename(scene), mi, d.meshes[mi].num_triangles, len(runs[0][0]), len(runs[0][1]), runs[0][2], min(r[3] for r in runs)), flush=True)
