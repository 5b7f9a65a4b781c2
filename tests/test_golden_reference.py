"""The oracle (and the host math it relies on) against golden vectors produced by the REAL reference
code: tests/golden/reference_vectors.json is written by oracle/ref_harness.cpp, which is linked against
the reference's own matrix.cpp / util.cpp / geometry.cpp / lights.cpp and headers (see
tests/golden/make_golden.sh).  These are the pins of the oracle; everything here runs on the CPU."""
import ctypes
import json
import os

import numpy as np
import pytest

import hexray_amd as hx
from desc_types import desc_of
from oracle_binding import dvec, oracle_lib

HERE = os.path.dirname(os.path.abspath(__file__))
TINY = os.path.join(HERE, "golden", "tiny.hexray")


@pytest.fixture(scope="module")
def golden():
    with open(os.path.join(HERE, "golden", "reference_vectors.json")) as f:
        return json.load(f)


def vec_text(v):
    return "(%s)" % ", ".join(repr(float(x)) for x in v)


def ops_text(ops):
    return "".join(f"\t{kind} {vec_text(v)}\n" for kind, v in ops)


def load_elements(text):
    """A scene made of `text` plus the mandatory camera."""
    return hx.HostScene(TINY, text)


def close(a, b, rel=1e-12, abs_=1e-12):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return np.all(np.abs(a - b) <= abs_ + rel * np.abs(b))


def test_transform_composition(golden):
    """Transform::scale/rotate/translate and the derived inverse / transposed inverse (reference src/matrix.cpp:119-152)."""
    text = ""
    for i, t in enumerate(golden["transforms"]):
        text += f"Node n{i} {{\n\tgeometry unit\n\tshader grey\n{ops_text(t['ops'])}}}\n"
    hs = load_elements(text)
    d = desc_of(hs)
    assert d.num_nodes == len(golden["transforms"])
    for i, t in enumerate(golden["transforms"]):
        T = d.nodes[i].T
        assert close(list(T.offset), t["offset"])
        assert close(list(T.m), t["m"])
        assert close(list(T.inv_m), t["inv"], rel=1e-11)
        assert close(list(T.tinv_m), t["tinv"], rel=1e-11)


def test_box_queries(golden):
    """BBox::testIntersect / closestIntersection (reference src/bbox.h:94-174): flags exact, distances to 1e-12."""
    lib = oracle_lib()
    for b in golden["boxes"]:
        near = ctypes.c_double()
        hit = lib.hxo_box_test(dvec(*b["lo"]), dvec(*b["hi"]), dvec(*b["o"]), dvec(*b["d"]), ctypes.byref(near))
        assert hit == b["test"], b
        if b["nearest"] >= 1e98:
            assert near.value >= 1e98
        else:
            assert close(near.value, b["nearest"], rel=1e-12, abs_=1e-13), b


def _check_hit(lib_fn, desc, index, case, out, rel=1e-10):
    ok = lib_fn(desc, index, dvec(*case["o"]), dvec(*case["d"]), out)
    assert ok == case["hit"], case
    if ok:
        assert close(out[0], case["dist"], rel=rel, abs_=1e-11), case
        assert close(out[1:4], case["ip"], rel=rel, abs_=1e-9), case
        assert close(out[4:7], case["norm"], rel=rel, abs_=1e-10), case
        assert close(out[7:9], [case["u"], case["v"]], rel=rel, abs_=1e-9), case


def test_analytic_primitives(golden):
    """Plane / Sphere / Cube::intersect in object space (reference src/geometry.cpp:31-147)."""
    lib = oracle_lib()
    text = ""
    for i, c in enumerate(golden["planes"]):
        text += f"Plane p{i} {{\n\ty {c['y']!r}\n\tlimit {c['limit']!r}\n}}\n"
    for i, c in enumerate(golden["spheres"]):
        text += f"Sphere s{i} {{\n\tO {vec_text(c['O'])}\n\tR {c['R']!r}\n\tuvscaling {c['uvscaling']!r}\n}}\n"
    for i, c in enumerate(golden["cubes"]):
        text += f"Cube c{i} {{\n\tO {vec_text(c['O'])}\n\tside {c['side']!r}\n}}\n"
    hs = load_elements(text)
    out = (ctypes.c_double * 12)()
    base = 1  # geometry 0 is tiny.hexray's own sphere
    for i, c in enumerate(golden["planes"]):
        _check_hit(lib.hxo_geometry_intersect, hs.desc, base + i, c, out)
    base += len(golden["planes"])
    for i, c in enumerate(golden["spheres"]):
        _check_hit(lib.hxo_geometry_intersect, hs.desc, base + i, c, out)
    base += len(golden["spheres"])
    for i, c in enumerate(golden["cubes"]):
        _check_hit(lib.hxo_geometry_intersect, hs.desc, base + i, c, out)


def test_csg(golden):
    """CSGUnion / CSGInter / CSGDiff over a sphere and a cube (reference src/geometry.cpp:149-194)."""
    lib = oracle_lib()
    text = ""
    for i, c in enumerate(golden["csg"]):
        text += (f"Sphere cs{i} {{\n\tO {vec_text(c['sphere_O'])}\n\tR {c['sphere_R']!r}\n}}\n"
                 f"Cube cc{i} {{\n\tO {vec_text(c['cube_O'])}\n\tside {c['cube_side']!r}\n}}\n"
                 f"{c['kind']} op{i} {{\n\tleft cs{i}\n\tright cc{i}\n}}\n")
    hs = load_elements(text)
    out = (ctypes.c_double * 12)()
    for i, c in enumerate(golden["csg"]):
        _check_hit(lib.hxo_geometry_intersect, hs.desc, 1 + 3 * i + 2, c, out, rel=1e-9)


def test_csg_nested(golden):
    """An operation over a sphere and a cube as the left / right operand of an outer operation (CSGBase::intersect is generic over Geometry*)."""
    lib = oracle_lib()
    text = ""
    for i, c in enumerate(golden["csg_nested"]):
        inner, other = f"in{i}", f"ns{i}"
        text += (f"Sphere is{i} {{\n\tO {vec_text(c['sphere_O'])}\n\tR {c['sphere_R']!r}\n}}\n"
                 f"Cube ic{i} {{\n\tO {vec_text(c['cube_O'])}\n\tside {c['cube_side']!r}\n}}\n"
                 f"{c['inner']} {inner} {{\n\tleft is{i}\n\tright ic{i}\n}}\n"
                 f"Sphere {other} {{\n\tO {vec_text(c['sphere2_O'])}\n\tR {c['sphere2_R']!r}\n}}\n"
                 f"{c['outer']} out{i} {{\n\tleft {inner if c['inner_is_left'] else other}\n\tright {other if c['inner_is_left'] else inner}\n}}\n")
    hs = load_elements(text)
    out = (ctypes.c_double * 12)()
    hits = 0
    for i, c in enumerate(golden["csg_nested"]):
        _check_hit(lib.hxo_geometry_intersect, hs.desc, 1 + 5 * i + 4, c, out, rel=1e-9)
        hits += c["hit"]
    assert hits > 200


def test_node_intersect(golden):
    """Node::intersect: ray into object space, hit back to world space (reference src/geometry.cpp:196-208)."""
    lib = oracle_lib()
    text = ""
    for i, c in enumerate(golden["nodes"]):
        if c["kind"] == "Plane":
            text += f"Plane g{i} {{\n\ty {c['a']!r}\n\tlimit {c['b']!r}\n}}\n"
        elif c["kind"] == "Sphere":
            text += f"Sphere g{i} {{\n\tR {c['a']!r}\n}}\n"
        else:
            text += f"Cube g{i} {{\n\tside {c['a']!r}\n}}\n"
        text += f"Node n{i} {{\n\tgeometry g{i}\n\tshader grey\n{ops_text(c['ops'])}}}\n"
    hs = load_elements(text)
    out = (ctypes.c_double * 12)()
    for i, c in enumerate(golden["nodes"]):
        _check_hit(lib.hxo_node_intersect, hs.desc, i, c, out, rel=1e-9)


def test_rect_light(golden):
    """RectLight::beginFrame / intersect / getSolidAngle / getNthSample (reference src/lights.cpp:32-102)."""
    lib = oracle_lib()
    text = ""
    for i, c in enumerate(golden["rect_lights"]):
        text += (f"RectLight l{i} {{\n{ops_text(c['ops'])}\tcolor {vec_text(c['color'])}\n\tpower {c['power']!r}\n"
                 f"\txSubd {c['xSubd']}\n\tySubd {c['ySubd']}\n}}\n")
    hs = load_elements(text)
    d = desc_of(hs)
    assert d.num_lights == len(golden["rect_lights"])
    for i, c in enumerate(golden["rect_lights"]):
        L = d.lights[i]
        assert close(L.scale_factor, c["scale_factor"], rel=1e-6)
        assert close([L.color[k] * L.power for k in range(3)], c["light_color"], rel=1e-6)
        dist = ctypes.c_double(c["dist_in"])
        side = lib.hxo_light_intersect(hs.desc, i, dvec(*c["o"]), dvec(*c["d"]), ctypes.byref(dist))
        assert side == c["side"], c
        assert close(dist.value, c["dist_out"], rel=1e-10)
        assert close(lib.hxo_light_solid_angle(hs.desc, i, dvec(*c["shade"])), c["solid_angle"], rel=1e-10, abs_=1e-15)
        pos, col = (ctypes.c_double * 3)(), (ctypes.c_float * 3)()
        lib.hxo_light_sample(hs.desc, i, c["idx"], dvec(*c["shade"]), c["j1"], c["j2"], pos, col)
        assert close(list(pos), c["sample_pos"], rel=1e-10, abs_=1e-9)
        assert close(list(col), c["sample_color"], rel=1e-5, abs_=1e-7)


def test_srgb_table(golden):
    """The 4097-entry sRGB table (reference src/sdl.cpp:404-419 over src/color.h:36-47) and decompress_sRGB."""
    lib = hx.host_lib()
    table = golden["srgb_table"]
    assert len(table) == 4097
    for i in range(1, 4096):
        # x*4096 truncates back to i for x = i/4096 (exact in float)
        assert lib.hxh_srgb8(ctypes.c_float(i / 4096.0)) == table[i], i
    assert lib.hxh_srgb8(ctypes.c_float(0.0)) == 0 and lib.hxh_srgb8(ctypes.c_float(1.5)) == 255
