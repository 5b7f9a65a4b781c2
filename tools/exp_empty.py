import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import hexray_amd as hx
hx.init(0)
base = os.path.join(ROOT, "tests", "golden", "tiny.hexray")
env = "CubemapEnvironment environment {\n assumedGamma 2.2\n folder \"../../data/env/mountain\"\n}\n"
for name, extra, kw in (("empty-noenv-1spp", "", dict(width=1920, height=1080, want_aa=False)),
                        ("empty-noenv-16spp-dof", "", dict(width=1920, height=1080, dof=True, num_samples=16)),
                        ("empty-env-16spp-dof", env, dict(width=1920, height=1080, dof=True, num_samples=16)),
                        ("empty-env-16spp-nodof(gi-off,MC)", env, dict(width=1920, height=1080))):
    hs = hx.HostScene(base, hx.overlay_text(**kw) + extra)
    ds = hx.DeviceScene(hs)
    p = hx.frame_params(tile_size=32, mode=hx.MODE_MONTE_CARLO, spp=16) if "nodof" in name else hx.frame_params(tile_size=32)
    ms = []
    for i in range(3):
        ds.render(p); ms.append(ds.last_kernel_ms())
    print(name, "kernel ms %.3f" % min(ms))
    ds.close()
