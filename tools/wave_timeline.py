"""Developer aid (a -DHXR_WAVE_TIMELINE build, tools/build_alt.sh wt -DHXR_WAVE_TIMELINE; HXR_HIP_LIB=hexray_amd/lib/libhexray_hip_wt.so): when the waves of
the headline frame's render kernel start and leave, and their longest packets -- steady state against the first frame of a fresh handle.
usage: HXR_HIP_LIB=... python3 tools/wave_timeline.py"""
import os, sys, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import hexray_amd as hx
hx.init(0)
if os.environ.get("HXR_WL"):   # another bench workload (steady state and first frame only), e.g. HXR_WL=meshes1
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py")); bench = importlib.util.module_from_spec(spec); spec.loader.exec_module(bench)
    scene_name, ov, desc = bench.WORKLOADS[os.environ["HXR_WL"]]
    hs = hx.HostScene(os.path.join(ROOT, "data", scene_name), hx.overlay_text(**ov))
else:
    hs = hx.HostScene(os.path.join(ROOT, "data", "meshes.hexray"), hx.overlay_text(width=1920, height=1080, dof=True, num_samples=16))
st = torch.cuda.current_stream().cuda_stream
frame = torch.zeros((1080, 1920, 3), dtype=torch.float32, device="cuda")
p = hx.frame_params(tile_size=64)
lib = hx.hip_lib()
lib.hxr_debug_wave_timeline.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t]
WAVES = 16384   # 8192 slots of the main kernel, then those of the replicating blocks launched beside it
def timeline(ds, label):
    both = np.zeros((WAVES, 4), dtype=np.uint64)
    rc = lib.hxr_debug_wave_timeline(ds._s, both.ctypes.data, WAVES)
    assert rc == 0
    t0_all = both[both[:, 1] > 0][:, 0].min()
    for name, w in (("", both[:8192]), (" -- the replicating blocks", both[8192:])):
        w = w[w[:, 1] > 0]
        if len(w): one_kernel(ds, label + name, w, t0_all)
def one_kernel(ds, label, w, t0):
    start = (w[:, 0] - t0).astype(np.float64) / 100.0   # us
    end = (w[:, 1] - t0).astype(np.float64) / 100.0
    longest = w[:, 2].astype(np.float64) / 100.0
    longest_at = ((w[:, 3] & np.uint64((1 << 48) - 1)) - t0).astype(np.float64) / 100.0
    packets = ((w[:, 3] >> np.uint64(48)) & np.uint64(0xfff)).astype(np.int64)
    kind = (w[:, 3] >> np.uint64(60)).astype(np.int64)   # of the wave's longest packet: 1 listed, 2 one pixel, 4 sample part, 8 from the front of the list
    kinds = {0: "chunk", 1: "listed chunk", 9: "front chunk", 3: "pixel (back)", 11: "pixel", 5: "part (back)", 13: "part"}
    q = lambda a, f: np.percentile(a, f)
    print("%s: kernel %.3f ms; %d waves; starts by %.0f us (p99); a wave leaves at p10 %.0f  p50 %.0f  p90 %.0f  p99 %.0f  max %.0f us; packets per wave p50 %d (min %d, max %d)" % (
        label, ds.last_kernel_ms(), len(w), q(start, 99), q(end, 10), q(end, 50), q(end, 90), q(end, 99), end.max(), q(packets, 50), packets.min(), packets.max()))
    order = np.argsort(-longest)[:12]
    print("   longest packets (us, began at us): " + "  ".join("%.0f@%.0f %s" % (longest[i], longest_at[i], kinds.get(int(kind[i]), str(int(kind[i])))) for i in order))
    late = np.argsort(-end)[:8]
    print("   last waves to leave (left at, their longest packet@began): " + "  ".join("%.0f:%.0f@%.0f" % (end[i], longest[i], longest_at[i]) for i in late), flush=True)
def fresh(env, label, n=2):
    for k, v in env.items(): os.environ[k] = v
    for _ in range(n):
        ds = hx.DeviceScene(hs)
        ds.render_device(p, frame.data_ptr(), 1920 * 12, st); torch.cuda.synchronize()
        timeline(ds, "first frame, " + label)
        ds.close()
    for k in env: del os.environ[k]
if os.environ.get("HXR_SHARD"):   # "rank,count[,tile]": the steady state of one shard of the frame only, e.g. HXR_SHARD=0,8
    v = [int(x) for x in os.environ["HXR_SHARD"].split(",")]
    ps = hx.frame_params(tile_size=v[2] if len(v) > 2 else 8, shard_rank=v[0], shard_count=v[1], layout=hx.LAYOUT_TILES)
    ds = hx.DeviceScene(hs, accel=hx.ACCEL_SAH if os.environ.get("HXR_ACCEL") == "sah" else hx.ACCEL_REFERENCE)
    buf = torch.zeros(ds.output_bytes(ps, 0) // 4 + 16, dtype=torch.float32, device="cuda")
    for i in range(7):
        ds.render_device(ps, buf.data_ptr(), 0, st); torch.cuda.synchronize()
        if i >= 4: timeline(ds, "shard %s, steady state" % os.environ["HXR_SHARD"])
    ds.close()
    sys.exit(0)
ds = hx.DeviceScene(hs)
for i in range(6):
    ds.render_device(p, frame.data_ptr(), 1920 * 12, st); torch.cuda.synchronize()
    if i >= 4: timeline(ds, "steady state")
ds.close()
fresh({}, "default (rectangle chunks listed, at the raised priority)")
fresh({"HXR_NO_FIRST_PRIORITY": "1"}, "listed, no priority")
fresh({"HXR_SCHED_NO_RECTS": "1"}, "nothing listed")
