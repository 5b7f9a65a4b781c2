"""Round 4, VERDICT r3 #3: the first frame of a fresh handle (no cost map) under several settings, against the steady state.  usage: python3 tools/r4_first_frame.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import hexray_amd as hx
hx.init(0)
hs = hx.HostScene(os.path.join(ROOT, "data", "meshes.hexray"), hx.overlay_text(width=1920, height=1080, dof=True, num_samples=16))
st = torch.cuda.current_stream().cuda_stream
frame = torch.zeros((1080, 1920, 3), dtype=torch.float32, device="cuda")
ref = None
p = hx.frame_params(tile_size=64)
def first(env, n=5):
    for k, v in env.items(): os.environ[k] = v
    ms = []
    for _ in range(n):
        ds = hx.DeviceScene(hs)
        frame.zero_()
        ds.render_device(p, frame.data_ptr(), 1920 * 12, st); torch.cuda.synchronize()
        ms.append(ds.last_kernel_ms())
        assert torch.equal(frame, ref), "a first frame differs from the steady-state frame under %r" % (env,)
        t = ds.frame_timing()
        ds.close()
    for k in env: del os.environ[k]
    return np.median(ms), min(ms), t
ds = hx.DeviceScene(hs)
steady = []
for i in range(8):
    ds.render_device(p, frame.data_ptr(), 1920 * 12, st); torch.cuda.synchronize()
    if i >= 3: steady.append(ds.last_kernel_ms())
print("steady %.3f ms" % np.mean(steady), ds.frame_timing(), flush=True)
ref = frame.clone()
for name, env in (("default", {}), ("3-waves instance", {"HXR_MESH_MC_FOUR": "0"}), ("no rects", {"HXR_SCHED_NO_RECTS": "1"}), ("3 waves, no rects", {"HXR_MESH_MC_FOUR": "0", "HXR_SCHED_NO_RECTS": "1"})):
    med, mn, t = first(env)
    print("first frame, %-20s median %.3f min %.3f ms   listed front %d back %d" % (name, med, mn, t["sched_front_items"], t["sched_back_items"]), flush=True)
