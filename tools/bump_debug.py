"""Developer aid: one bumped node of each geometry kind, GPU against oracle."""
import os, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import hexray_amd as hx
from oracle_binding import oracle_render
hx.init(0)
d = tempfile.mkdtemp()
data = os.path.relpath(os.path.join(ROOT, "data"), d)
HEAD = """GlobalSettings {\n frameWidth 80\n frameHeight 60\n ambientLight (0.2,0.2,0.2)\n wantAA off\n prepassSamples 0\n}
Camera camera {\n pos (0, 20, -50)\n aspectRatio 1.3333\n yaw 0\n pitch -15\n roll 0\n fov 70\n}
PointLight pl {\n pos (30, 80, -40)\n power 30000\n}
Lambert lam {\n color (0.8, 0.7, 0.6)\n}
BumpTexture bump_tex {\n file "%s/texture/zar-bump.bmp"\n strength 4\n}
Bumps bump_proc {\n strength 0.2\n}
Sphere sp {\n R 8\n}\nCube cu {\n side 12\n}\nPlane pq {\n y 0\n limit 12\n}\nSphere sp2 {\n O (3, 2, 0)\n R 7\n}\nCSGDiff cs {\n left cu\n right sp2\n}
Mesh me {\n file "%s/geom/truncated_cube.obj"\n faceted true\n}
""" % (data, data)
for geom in ("sp", "cu", "pq", "cs", "me"):
    for bump in ("bump_tex", "bump_proc"):
        for tr in ("", " rotate (30, 20, 10)\n scale (1.2, 0.8, 1.5)\n"):
            text = HEAD + "Node n {\n geometry %s\n shader lam\n bump %s\n%s translate (0, 10, 0)\n%s}\n" % (geom, bump, tr, " scale (5,5,5)\n" if geom == "me" and not tr else "")
            p = os.path.join(d, "s.hexray"); open(p, "w").write(text)
            hs = hx.HostScene(p)
            ds = hx.DeviceScene(hs)
            got = ds.render(); want, _ = oracle_render(hs)
            rms = np.sqrt(np.mean((got.astype(np.float64) - want) ** 2, axis=(0, 1))).max()
            print(geom, bump, "transformed" if tr else "plain", "rms %.2e" % rms, "nan gpu/oracle", int(np.isnan(got).sum()), int(np.isnan(want).sum()), flush=True)
