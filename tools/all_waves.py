"""Developer aid: the catch-all kernels at 2 and 3 waves per SIMD (HXR_ALL_WAVES) on scenes that need them."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import hexray_amd as hx
hx.init(0)
cube = "Cube extra_cube {\n side 8\n}\nLambert extra_l {\n color (0.7,0.7,0.7)\n}\nNode extra_n {\n geometry extra_cube\n shader extra_l\n translate (50, 10, 80)\n}\n"
cases = [("smallpt.hexray + cube, 16 paths", "smallpt.hexray", hx.overlay_text(1920, 1080, num_paths=16, extra=cube)),
         ("boxed.hexray 1 spp", "boxed.hexray", hx.overlay_text(1920, 1080, want_aa=False)),
         ("hw10/bokeh.hexray 8 spp", "hw10/bokeh.hexray", hx.overlay_text(1920, 1080, num_samples=8)),
         ("kdtree_test 16 spp", "kdtree_test.hexray", hx.overlay_text(1920, 1080, dof=True, num_samples=16)),
         ("cornell box + cube 8 paths", "cornell_box.hexray", hx.overlay_text(1920, 1080, num_paths=8, extra=cube))]
for name, scene, ov in cases:
    hs = hx.HostScene(os.path.join(ROOT, "data", scene), ov)
    for waves in ("2", "3"):
        os.environ["HXR_ALL_WAVES"] = waves
        ds = hx.DeviceScene(hs)
        ms = []
        for i in range(4):
            ds.render(hx.frame_params(tile_size=32)); ms.append(ds.last_kernel_ms())
        print("%-34s %s waves/SIMD: %s" % (name, waves, " ".join("%.2f" % m for m in ms)), flush=True)
        ds.close()
