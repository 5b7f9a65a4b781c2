"""Developer aid: where is the latency floor of the headline frame?  Finds the slowest 32x32 tile (each rendered alone as a
one-tile shard), then its slowest 8x8 tile, and prints time and per-ray work counters of that tile alone (16 packets on an empty GPU:
the kernel time IS the longest packet).  With a -DHXR_PHASE_TIMERS library (HXR_HIP_LIB) and HXR_TILE8=<index> it prints the phase split."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import numpy as np
import hexray_amd as hx
hx.init(0)
hs = hx.HostScene(os.path.join(ROOT, "data", "meshes.hexray"), hx.overlay_text(width=1920, height=1080, dof=True, num_samples=int(os.environ.get("SPP", "16"))))
ds = hx.DeviceScene(hs, adaptive_schedule=False, accel=hx.ACCEL_SAH if os.environ.get("HXR_ACCEL") == "sah" else hx.ACCEL_REFERENCE)
st = torch.cuda.current_stream().cuda_stream
buf = torch.zeros(32 * 32 * 3 + 64, dtype=torch.float32, device="cuda")


def tile_ms(tile, index, count, counters=False, reps=2):
    p = hx.frame_params(tile_size=tile, shard_rank=index, shard_count=count, layout=hx.LAYOUT_TILES, collect_counters=counters)
    best = 1e9
    for _ in range(reps):
        ds.render_device(p, buf.data_ptr(), 0, st)
        torch.cuda.synchronize()
        best = min(best, ds.last_kernel_ms())
    return best


if os.environ.get("HXR_TILE8"):
    t8 = int(os.environ["HXR_TILE8"])
    ms = tile_ms(8, t8, 240 * 135)
    ph = list(ds.counters_raw().reserved_)[:6]
    names = os.environ.get("HXR_PHASE_NAMES", "dispense,gen(+reduce),query(nodes),mesh walk,leaf,shade").split(",")
    tot = sum(ph) or 1
    print("8x8 tile %d: kernel %.3f ms; wave-clock shares:" % (t8, ms), "  ".join("%s %.1f%%" % (n, 100.0 * v / tot) for n, v in zip(names, ph)), " raw clocks", ph, flush=True)
    if os.environ.get("HXR_CHAIN_COUNTS"):   # the same tile through the counting build of the SAME library: work per phase
        tile_ms(8, t8, 240 * 135, counters=True, reps=1)
        c = ds.counters()
        print("  counters of that tile: samples %d raycasts %d shadow %d kd_visits %d bbox %d tri_tests %d" % (c["camera_samples"], c["raycasts"], c["shadow_queries"], c["kd_visits"], c["bbox_tests"], c["tri_tests"]), flush=True)
    sys.exit(0)

times = np.array([tile_ms(32, t, 60 * 34, reps=1) for t in range(60 * 34)])
order = np.argsort(times)[::-1]
print("slowest 32x32 tiles (index, x, y, ms):", [(int(t), int(t % 60) * 32, int(t // 60) * 32, round(float(times[t]), 3)) for t in order[:6]], "median %.3f" % np.median(times), flush=True)
t = int(order[0])
x0, y0 = (t % 60) * 32, (t // 60) * 32
sub = []
for by in range(4):
    for bx in range(4):
        t8 = ((y0 // 8) + by) * 240 + (x0 // 8) + bx
        sub.append((tile_ms(8, t8, 240 * 135), t8))
sub.sort(reverse=True)
print("its 8x8 tiles (ms, index):", [(round(m, 3), i) for m, i in sub[:5]], flush=True)
ms, t8 = sub[0]
tile_ms(8, t8, 240 * 135, counters=True, reps=1)
c = ds.counters()
n = c["camera_samples"]
print("worst 8x8 tile %d: %.3f ms for %d samples; per sample: raycasts %.2f shadow %.2f node tests %.1f kd visits %.1f box tests %.1f tri tests %.1f" %
      (t8, ms, n, c["raycasts"] / n, c["shadow_queries"] / n, c["node_tests"] / n, c["kd_visits"] / n, c["bbox_tests"] / n, c["tri_tests"] / n), flush=True)
print("HXR_TILE8=%d" % t8)
