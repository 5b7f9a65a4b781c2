import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import hexray_amd as hx
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
hx.init(0)
for name, ov in (("simple.hexray", dict(width=1920, height=1080, want_aa=False)), ("meshes.hexray", dict(width=1920, height=1080, dof=True, num_samples=16)), ("meshes.hexray", dict(width=1920, height=1080)), ("kdtree_test.hexray", dict(width=960, height=540)), ("cornell_box.hexray", dict(width=480, height=480, num_paths=4))):
    hs = hx.HostScene(os.path.join(ROOT, "data", name), hx.overlay_text(**ov))
    ds = hx.DeviceScene(hs)
    ds.render(hx.frame_params(collect_counters=True))
    c = ds.counters()
    print(name, ov, "bbox_tests", c["bbox_tests"], "kd_visits", c["kd_visits"], "box_test_mismatch", c["box_test_mismatch"], flush=True)
