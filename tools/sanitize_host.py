"""Developer aid (CPU only): the host loader, the oracle and the image I/O under AddressSanitizer + UBSan, over the shipped scenes and 40
generated ones (tools/sanitize_host.sh builds the two libraries with -fsanitize=address,undefined into a scratch directory and runs this)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import hexray_amd as hx
hx.HOST_LIB_PATH = os.path.join(os.environ['HXR_SANITIZED_DIR'], 'libhexray_host.so')
import oracle_binding
oracle_binding.ORACLE_LIB_PATH = os.path.join(os.environ['HXR_SANITIZED_DIR'], 'liboracle.so')
import numpy as np
from oracle_binding import oracle_render
import importlib.util
spec = importlib.util.spec_from_file_location("t", os.path.join(ROOT, "tests", "test_gpu_fuzz.py")); t = importlib.util.module_from_spec(spec); spec.loader.exec_module(t)
import tempfile
d = tempfile.mkdtemp()
# the shipped scenes at a small size, every mode
for scene, ov in (("meshes.hexray", hx.overlay_text(96, 72)), ("meshes.hexray", hx.overlay_text(96, 72, dof=True, num_samples=2)), ("heightfield.hexray", hx.overlay_text(96, 72, want_aa=False)),
                  ("smallpt.hexray", hx.overlay_text(64, 48, num_paths=2)), ("kdtree_test.hexray", hx.overlay_text(96, 72)), ("cornell_box.hexray", hx.overlay_text(64, 48, num_paths=2)),
                  ("boxed.hexray", hx.overlay_text(64, 48, want_aa=False)), ("hw10/bokeh.hexray", hx.overlay_text(64, 48, num_samples=2)), ("beer.hexray", hx.overlay_text(64, 48, want_aa=False))):
    for kw in ({}, {"gpu_kd_build": True, "gpu_heightfield_blur": True, "gpu_prepare_triangles": True}):
        hs = hx.HostScene(os.path.join(ROOT, "data", scene), ov, **kw)
        if not kw:
            out, c = oracle_render(hs, hx.frame_params(collect_counters=True))
            assert np.isfinite(out).all()
        hs.close()
    print("ok", scene, flush=True)
# generated scenes
for seed in range(40):
    text, mode = t.make_scene(seed, os.path.relpath(os.path.join(ROOT, "data", "geom"), d))
    p = os.path.join(d, "s.hexray"); open(p, "w").write(text)
    hs = hx.HostScene(p)
    for fp in ({}, dict(mode=hx.MODE_PREPASS, spp=2), dict(mode=hx.MODE_SINGLE, foveated_radius=12, vip=(20, 20))):
        out, c = oracle_render(hs, hx.frame_params(**fp))
        assert np.isfinite(out).all()
    hs.close()
print("fuzz ok")
# image writers / readers
a = np.random.rand(17, 23, 3).astype(np.float32)
hx.save_bmp(os.path.join(d, "x.bmp"), a); hx.save_exr(os.path.join(d, "x.exr"), a); hx.load_image(os.path.join(d, "x.bmp")); hx.load_image(os.path.join(d, "x.exr"))
print("io ok")
# the scene parser on damaged input: 400 random mutations of shipped scene files either load or are refused with a message
import random
r = random.Random(7)
work = tempfile.mkdtemp()
for sub in ("geom", "texture", "env"):
    os.symlink(os.path.join(ROOT, "data", sub), os.path.join(work, sub))
srcs = [open(os.path.join(ROOT, "data", n)).read() for n in ("meshes.hexray", "kdtree_test.hexray", "smallpt.hexray", "heightfield.hexray")]
loaded = refused = 0
for i in range(400):
    b = list(r.choice(srcs))
    for k in range(r.randint(1, 6)):
        op, pos = r.random(), r.randrange(len(b))
        if op < 0.4:
            del b[pos:pos + r.randint(1, 12)]
        elif op < 0.7:
            b[pos:pos] = list(r.choice(["{", "}", "(", ")", "\n", " 1e999 ", " -5 ", "\"", "file \"nope.obj\"", "//", "/*", "Mesh x {", "999999999999"]))
        else:
            b[pos] = chr(r.randint(32, 126))
    p = os.path.join(work, "mutated.hexray")
    open(p, "w").write("".join(b))
    try:
        hx.HostScene(p).close()
        loaded += 1
    except ValueError:
        refused += 1
print("mutated scene files: %d loaded, %d refused, no crash" % (loaded, refused))
