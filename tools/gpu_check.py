"""Developer aid: render a scene on the GPU and with the CPU oracle, print RMS / counter differences."""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import hexray_amd as hx
from oracle_binding import oracle_render

def main():
    scene, w, h, mode, spp = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
    extra = sys.argv[6].replace("\\n", "\n") if len(sys.argv) > 6 else ""
    tag = os.path.splitext(os.path.basename(scene))[0] + f"_{w}x{h}_m{mode}_s{spp}"
    hs = hx.HostScene(scene, hx.overlay_text(width=w, height=h) + extra)
    hx.init(0)
    ds = hx.DeviceScene(hs)
    p = hx.frame_params(mode=mode, spp=spp, collect_counters=True)
    t = time.time(); g = ds.render(p); tg = time.time() - t
    gc = ds.counters()
    p2 = hx.frame_params(mode=mode, spp=spp)
    ds.render(p2); t = time.time(); ds.render(p2); tg2 = time.time() - t
    kms = ds.last_kernel_ms()
    t = time.time(); o, oc = oracle_render(hs, hx.frame_params(mode=mode, spp=spp)); to = time.time() - t
    diff = g.astype(np.float64) - o
    rms = np.sqrt((diff ** 2).mean(axis=(0, 1)))
    print(tag, "gpu %.3fs (counted) %.3fs (plain, kernel %.2f ms)  oracle %.3fs" % (tg, tg2, kms, to))
    print("  per-channel RMS", rms, "max abs", np.abs(diff).max(), "pixels>1e-3:", int((np.abs(diff).max(axis=2) > 1e-3).sum()))
    bad = {k: (gc[k], oc[k]) for k in gc if gc[k] != oc[k]}
    print("  counters equal" if not bad else "  COUNTER MISMATCH %s" % bad)
    print("  samples/s gpu(kernel) %.1f M" % (oc["camera_samples"] / (kms / 1e3) / 1e6))
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    hx.save_bmp(os.path.join(ROOT, "gpurun_out", tag + "_gpu.bmp"), g)
    np.save(os.path.join(ROOT, "gpurun_out", tag + "_diff.npy"), np.abs(diff).max(axis=2).astype(np.float32))

main()
