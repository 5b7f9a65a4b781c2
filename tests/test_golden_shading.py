"""The oracle's shaders, textures and samplers against golden vectors produced by the REAL reference code:
tests/golden/reference_shading_vectors.json is written by oracle/ref_harness_shading.cpp, which is linked against the reference's
own shading.cpp / util.cpp (see its header for what can and what cannot be linked in this image).  Shaders ask the harness's
recording raytrace() for their secondary rays; the oracle's probe mode (hxo_shade_probe) records the same way and answers with
the same colour function, so a case compares both the rays asked for and the colour made of the answers.  CPU only."""
import ctypes
import json
import os

import numpy as np
import pytest

import hexray_amd as hx
from desc_types import desc_of
from oracle_binding import ProbeHit, dvec, oracle_lib

HERE = os.path.dirname(os.path.abspath(__file__))
TINY = os.path.join(HERE, "golden", "tiny.hexray")
F32 = 3e-6    # colours are float arithmetic on both sides (the oracle is built with the reference's Release flags, the harness with -O2)
F64 = 1e-11


@pytest.fixture(scope="module")
def golden():
    with open(os.path.join(HERE, "golden", "reference_shading_vectors.json")) as f:
        return json.load(f)


def close(a, b, rel, abs_=0.0):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return bool(np.all(np.abs(a - b) <= abs_ + rel * np.abs(b)))


def vec_text(v):
    return "(%s)" % ", ".join(repr(float(x)) for x in v)


def hit_of(c):
    h = ProbeHit()
    for name in ("ip", "norm", "dNdx", "dNdy"):
        for k in range(3):
            getattr(h, name)[k] = c[name][k]
    h.u, h.v = c["u"], c["v"]
    return h


def shade(lib, desc, shader, c, max_rays=64):
    out = (ctypes.c_float * 3)()
    rays = (ctypes.c_double * (8 * max_rays))()
    n = lib.hxo_shade_probe(desc, shader, dvec(*c["o"]), dvec(*c["d"]), c["depth"], c["flags"], ctypes.byref(hit_of(c)), out, rays, max_rays)
    got = [dict(o=list(rays[8 * i:8 * i + 3]), d=list(rays[8 * i + 3:8 * i + 6]), depth=int(rays[8 * i + 6]), flags=int(rays[8 * i + 7])) for i in range(min(n, max_rays))]
    return list(out), n, got


def same_rays(got, want, rel=F64, abs_=1e-12):
    assert len(got) == len(want), (len(got), len(want))
    for g, w in zip(got, want):
        assert g["depth"] == w["depth"] and g["flags"] == w["flags"], (g, w)
        assert close(g["o"], w["o"], rel, 1e-11), (g, w)
        assert close(g["d"], w["d"], rel, abs_), (g, w)


def test_box_split(golden):
    """BBox::split (reference src/bbox.h:210-216): the child boxes the oracle's kd descent carries."""
    lib = oracle_lib()
    out = (ctypes.c_double * 12)()
    for c in golden["box_split"]:
        lib.hxo_box_split(dvec(*c["lo"]), dvec(*c["hi"]), c["axis"], c["where"], out)
        assert list(out) == c["left_lo"] + c["left_hi"] + c["right_lo"] + c["right_hi"]   # copies of the inputs: exact


def test_samplers_against_the_reference_generator_stream(golden):
    """unitDiskSample / hemisphereSample (reference src/util.cpp:101-129) and the glossy loop of Reflection::computeColor
    (src/shading.cpp:165-184), driven by the very numbers the reference's mt19937 handed out: the oracle's maps from uniform
    draws to disk points, hemisphere directions and perturbed mirror directions reproduce the reference's outputs."""
    lib = oracle_lib()
    stream = iter(golden["rng_stream"])
    xy = (ctypes.c_double * 2)()

    def disk():
        while True:
            u, v = next(stream), next(stream)
            if lib.hxo_disk_map(u, v, xy):
                return xy[0], xy[1]

    for want in golden["unit_disk"]:
        assert list(disk()) == want                                      # 2u-1 is exact: bit-equal
    out = (ctypes.c_double * 3)()
    for c in golden["hemisphere"]:
        u, v = next(stream), next(stream)
        lib.hxo_hemisphere(u, v, dvec(*c["n"]), out)
        assert close(list(out), c["dir"], 1e-12, 1e-15), c
    for c in golden["glossy"]:
        n = np.array(c["norm"]) * (1.0 if np.dot(c["d"], c["norm"]) < 0 else -1.0)   # faceforward
        assert len(c["rays"]) == c["num_samples"]
        for ray in c["rays"]:
            while True:                                                  # `do { ... } while (dot(reflected, n) < 0)`
                x, y = disk()
                lib.hxo_glossy_dir(dvec(*n), dvec(*c["d"]), c["glossiness"], x, y, out)
                if not np.dot(list(out), n) < 0:
                    break
            assert close(list(out), ray["d"], 1e-7, 1e-7), (c, ray)       # the perturbation's scale, pow(10.0f, 2 - 8 * glossiness), is a float
            assert ray["depth"] == c["depth"] + 1
            assert close(ray["o"], np.array(c["ip"]) + n * 1e-6, 1e-12, 1e-12)


def test_textures(golden):
    """CheckerTexture::sample, Fresnel::sample, Bumps::modifyNormal (reference src/shading.cpp:50-55,326-343,363-377)."""
    lib = oracle_lib()
    text = ""
    for i, c in enumerate(golden["checker"]):
        text += f"CheckerTexture ck{i} {{\n\tcolor1 {vec_text(c['color1'])}\n\tcolor2 {vec_text(c['color2'])}\n\tscaling {c['scaling']!r}\n}}\n"
    for i, c in enumerate(golden["fresnel"]):
        text += f"Fresnel fr{i} {{\n\tior {c['ior']!r}\n}}\n"
    for i, c in enumerate(golden["bumps"]):
        text += f"Bumps bm{i} {{\n\tstrength {c['strength']!r}\n}}\n"
    hs = hx.HostScene(TINY, text)
    d = desc_of(hs)
    assert d.num_textures == len(golden["checker"]) + len(golden["fresnel"]) + len(golden["bumps"])
    out = (ctypes.c_float * 3)()
    for i, c in enumerate(golden["checker"]):
        lib.hxo_texture_sample(hs.desc, i, dvec(*c["d"]), ctypes.byref(hit_of(c)), out)
        assert list(out) == c["out"], c                                  # one of two colours: exact
    base = len(golden["checker"])
    for i, c in enumerate(golden["fresnel"]):
        lib.hxo_texture_sample(hs.desc, base + i, dvec(*c["d"]), ctypes.byref(hit_of(c)), out)
        assert close(list(out), c["out"], F32), c
    base += len(golden["fresnel"])
    for i, c in enumerate(golden["bumps"]):
        h = hit_of(c)
        lib.hxo_texture_bump(hs.desc, base + i, ctypes.byref(h))
        assert close(list(h.norm), c["norm_out"], 1e-9, 1e-12), c


def test_reflection_and_refraction(golden):
    """Reflection::computeColor (mirror branch) and Refraction::computeColor with refract() (reference src/shading.cpp:159-232):
    the secondary ray each sends to raytrace() -- origin offset, direction, depth, flags -- and the colour made of the answer;
    refraction cases include total internal reflection (no ray, black)."""
    lib = oracle_lib()
    text = ""
    for i, c in enumerate(golden["reflection"]):
        text += f"Reflection rl{i} {{\n\treflColor {vec_text(c['color'])}\n}}\n"
    for i, c in enumerate(golden["refraction"]):
        text += f"Refraction rr{i} {{\n\trefrColor {vec_text(c['color'])}\n\tior {c['ior']!r}\n}}\n"
    hs = hx.HostScene(TINY, text)
    assert desc_of(hs).num_shaders == 1 + len(golden["reflection"]) + len(golden["refraction"])
    for i, c in enumerate(golden["reflection"]):
        out, n, rays = shade(lib, hs.desc, 1 + i, c)
        same_rays(rays, c["rays"])
        assert n == 1 and close(out, c["out"], F32), c
    base = 1 + len(golden["reflection"])
    blocked = 0
    for i, c in enumerate(golden["refraction"]):
        out, n, rays = shade(lib, hs.desc, base + i, c)
        same_rays(rays, c["rays"], rel=1e-6, abs_=1e-7)                  # refract() works in float (src/shading.cpp:208-214)
        assert n == len(c["rays"]) and close(out, c["out"], 1e-5, 1e-6), c
        blocked += n == 0
    assert 10 < blocked < len(golden["refraction"]) // 2                 # total internal reflection is among the cases


def test_layered(golden):
    """Layered::computeColor (reference src/shading.cpp:315-324) over Const / Reflection / Refraction layers blended by a colour,
    a CheckerTexture or a Fresnel texture: the glass and the reflective coats of the config scenes are of this shape."""
    lib = oracle_lib()
    text = ""
    for i, c in enumerate(golden["layered"]):
        text += (f"CheckerTexture ck{i} {{\n\tcolor1 {vec_text(c['checker1'])}\n\tcolor2 {vec_text(c['checker2'])}\n\tscaling {c['checker_scaling']!r}\n}}\n"
                 f"Fresnel fr{i} {{\n\tior {c['ior']!r}\n}}\n"
                 f"Const k{i} {{\n\tcolor {vec_text(c['const'])}\n}}\n"
                 f"Reflection rl{i} {{\n\treflColor {vec_text(c['refl'])}\n}}\n"
                 f"Refraction rr{i} {{\n\trefrColor {vec_text(c['refr'])}\n\tior {c['ior']!r}\n}}\n"
                 f"Layered ly{i} {{\n")
        for L in c["layers"]:
            shader = ("k", "rl", "rr")[L["shader"]] + str(i)
            tex = ("", f" ck{i}", f" fr{i}")[L["tex"]]
            text += f"\tlayer {shader} {vec_text(L['blend'])}{tex}\n"
        text += "}\n"
    hs = hx.HostScene(TINY, text)
    assert desc_of(hs).num_shaders == 1 + 4 * len(golden["layered"])
    for i, c in enumerate(golden["layered"]):
        out, n, rays = shade(lib, hs.desc, 1 + 4 * i + 3, c)
        same_rays(rays, c["rays"], rel=1e-6, abs_=1e-7)
        assert n == len(c["rays"]) and close(out, c["out"], 1e-5, 2e-6), c


def test_specular_brdfs(golden):
    """BRDF::eval / spawnRay of Reflection and Refraction as pathtrace() uses them (reference src/shading.cpp:189-205,234-263)."""
    lib = oracle_lib()
    text = ""
    for i, c in enumerate(golden["brdf"]):
        if c["kind"]:
            text += f"Refraction b{i} {{\n\trefrColor {vec_text(c['color'])}\n\tior {c['ior']!r}\n}}\n"
        else:
            text += f"Reflection b{i} {{\n\treflColor {vec_text(c['color'])}\n}}\n"
    hs = hx.HostScene(TINY, text)
    out8, col, pdf, ev = (ctypes.c_double * 8)(), (ctypes.c_float * 3)(), ctypes.c_float(), (ctypes.c_float * 3)()
    for i, c in enumerate(golden["brdf"]):
        h = hit_of(c)
        lib.hxo_brdf_spawn(hs.desc, 1 + i, ctypes.byref(h), dvec(*c["o"]), dvec(*c["d"]), c["depth"] + 1, c["flags"], out8, col, ctypes.byref(pdf))
        assert pdf.value == c["pdf"], c
        assert close(list(col), c["spawn_color"], F32), c
        if c["pdf"] > 0:
            assert close(list(out8[0:3]), c["spawn_o"], 1e-12, 1e-11) and close(list(out8[3:6]), c["spawn_d"], 1e-6, 1e-7), c
            assert int(out8[7]) == c["spawn_flags"], c
        for w_out, want in ((c["mirror"], c["eval_mirror"]), (c["other"], c["eval_other"])):
            lib.hxo_brdf_eval(hs.desc, 1 + i, ctypes.byref(h), dvec(*c["d"]), dvec(*w_out), ev)
            assert close(list(ev), want, F32), c


# ---------------------------------------------------------------------------------------------------------------------------------
# Lambert / Phong (round 3): tests/golden/reference_diffuse_vectors.json is written by `oracle/ref_harness_shading diffuse`, where the
# reference's own Lambert::computeColor / Phong::computeColor run over the reference's own PointLight / RectLight, a harness-owned
# `Scene scene` (lights + ambientLight) and a visible() that records its two points and answers from a per-case bit pattern.
@pytest.fixture(scope="module")
def diffuse_golden():
    with open(os.path.join(HERE, "golden", "reference_diffuse_vectors.json")) as f:
        return json.load(f)


def diffuse_scene_text(c, with_lights=True):
    text = ""
    if with_lights:
        text += f"GlobalSettings {{\n\tambientLight {vec_text(c['ambient'])}\n}}\n"
        for k, L in enumerate(c["lights"]):
            if L["rect"]:
                ops = "".join(f"\t{o['op']} {vec_text(o['v'])}\n" for o in L["ops"])
                text += f"RectLight l{k} {{\n{ops}\tcolor {vec_text(L['color'])}\n\tpower {L['power']!r}\n\txSubd {L['xSubd']}\n\tySubd {L['ySubd']}\n}}\n"
            else:
                text += f"PointLight l{k} {{\n\tpos {vec_text(L['pos'])}\n\tcolor {vec_text(L['color'])}\n\tpower {L['power']!r}\n}}\n"
    tex = ""
    if c["textured"]:
        text += f"CheckerTexture ck {{\n\tcolor1 {vec_text(c['checker1'])}\n\tcolor2 {vec_text(c['checker2'])}\n\tscaling {c['checker_scaling']!r}\n}}\n"
        tex = "\ttexture ck\n"
    if c.get("phong"):
        text += f"Phong sh {{\n\tcolor {vec_text(c['color'])}\n\tspecular {vec_text(c['specular'])}\n\texponent {c['exponent']!r}\n{tex}}}\n"
    else:
        text += f"Lambert sh {{\n\tcolor {vec_text(c['color'])}\n{tex}}}\n"
    return text


def test_lambert_and_phong_against_the_reference(diffuse_golden):
    """Lambert::computeColor / Phong::computeColor (reference src/shading.cpp:57-91,112-145): for every light and light sample the two
    points handed to visible() (light sample position -- PointLight::pos or RectLight::getNthSample driven by the reference generator's
    own draws -- and ip + norm * 1e-6), which samples are skipped (zero light colour, blocked), and the colour: diffuse + specular
    terms, the division by the sample count, the ambient term."""
    lib = oracle_lib()
    kinds = {"phong": 0, "lambert": 0, "rect": 0, "blocked": 0, "skipped_lights": 0}
    for c in diffuse_golden["diffuse"]:
        hs = hx.HostScene(TINY, diffuse_scene_text(c))
        d = desc_of(hs)
        assert d.num_lights == len(c["lights"]) and d.num_shaders == 2
        out = (ctypes.c_float * 3)()
        max_vis = 64
        pts = (ctypes.c_double * (6 * max_vis))()
        n_vis, used = ctypes.c_int(), ctypes.c_int()
        draws = (ctypes.c_double * max(1, len(c["draws"])))(*c["draws"])
        n_rays = lib.hxo_shade_probe2(hs.desc, 1, dvec(*c["o"]), dvec(*c["d"]), c["depth"], c["flags"], ctypes.byref(hit_of(c)), out, None, 0,
                                      int(c["pattern"]), pts, max_vis, ctypes.byref(n_vis), draws if c["draws"] else None, len(c["draws"]), ctypes.byref(used))
        assert n_rays == 0                                                   # diffuse shaders trace nothing
        assert n_vis.value == len(c["visible_calls"]), (n_vis.value, len(c["visible_calls"]), c["lights"])
        for k, w in enumerate(c["visible_calls"]):
            assert close(list(pts[6 * k:6 * k + 3]), w["a"], 1e-11, 1e-10), (k, c["lights"])          # light sample position
            assert close(list(pts[6 * k + 3:6 * k + 6]), w["b"], 1e-12, 1e-12), k                      # ip + norm * 1e-6
        # every draw is consumed only when visible() is reached or skipped by a zero colour: the reference draws in getNthSample, before either
        assert used.value == len(c["draws"])
        assert close(list(out), c["out"], 2e-5, 2e-6), (list(out), c["out"], c["lights"])
        kinds["phong" if c["phong"] else "lambert"] += 1
        kinds["rect"] += any(L["rect"] for L in c["lights"])
        kinds["blocked"] += int(c["pattern"]) != 2 ** 64 - 1
        samples = sum(L["xSubd"] * L["ySubd"] if L["rect"] else 1 for L in c["lights"])
        kinds["skipped_lights"] += samples != len(c["visible_calls"])
        hs.close()
    assert min(kinds.values()) > 10, kinds


def test_lambert_brdf_against_the_reference(diffuse_golden):
    """Lambert::eval / spawnRay (reference src/shading.cpp:93-109) as pathtrace() calls them: origin offset along the faceforwarded
    normal, the hemisphere direction made of the reference generator's own two draws, RF_GI_DIFFUSE, colour and pdf."""
    lib = oracle_lib()
    out8, col, pdf, ev = (ctypes.c_double * 8)(), (ctypes.c_float * 3)(), ctypes.c_float(), (ctypes.c_float * 3)()
    for c in diffuse_golden["diffuse_brdf"]:
        hs = hx.HostScene(TINY, diffuse_scene_text(c, with_lights=False))
        h = hit_of(c)
        lib.hxo_brdf_spawn2(hs.desc, 1, ctypes.byref(h), dvec(*c["o"]), dvec(*c["d"]), c["depth"] + 1, c["flags"], c["draw_u"], c["draw_v"], out8, col, ctypes.byref(pdf))
        assert pdf.value == c["pdf"], c
        assert close(list(out8[0:3]), c["spawn_o"], 1e-12, 1e-11) and close(list(out8[3:6]), c["spawn_d"], 1e-12, 1e-14), c
        assert int(out8[6]) == c["depth"] + 1 and int(out8[7]) == c["spawn_flags"], c
        assert close(list(col), c["spawn_color"], F32, 1e-9), c
        lib.hxo_brdf_eval(hs.desc, 1, ctypes.byref(h), dvec(*c["d"]), dvec(*c["other"]), ev)
        assert close(list(ev), c["eval_other"], F32, 1e-9), c
        hs.close()
