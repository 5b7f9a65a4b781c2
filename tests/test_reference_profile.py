"""Whole-pipeline pin of the oracle (and of the host loader / kd builder under it): the number of times
each reference function runs for one deterministic frame.

SURVEY.md Appendix C records the gprof call counts of the REFERENCE binary for `data/simple.hexray` at
1920x1080, 1 spp (a deterministic, RNG-free frame; the EXR cubemap was not loaded there either).  A
restatement that takes one different branch anywhere -- a kd child visited in another order, a shadow
ray blocked by another node, a different triangle accepted -- changes these integers.  The oracle must
reproduce them exactly."""
import ctypes

import hexray_amd as hx
from conftest import scene_path
from oracle_binding import oracle_render

# SURVEY.md Appendix C, column "simple 1 spp"
REFERENCE_CALLS = {
    "camera_samples": 2_073_600,
    "raycasts": 2_073_600,          # raytrace / raycast
    "shadow_queries": 1_463_276,    # visible
    "node_tests": 13_966_357,       # Node::intersect
    "tri_tests": 60_638_783,        # Mesh::intersectTriangle
    "tex_bilinear": 584_798,        # getFilteredPixel (bitmap texture + bump map; env not loaded)
}
REFERENCE_BBOX_TESTS = 65_626_566   # BBox::testIntersect, render + kd build


def test_oracle_reproduces_reference_call_counts():
    hs = hx.HostScene(scene_path("simple.hexray"), hx.overlay_text(1920, 1080, interactive=False))
    _, cnt = oracle_render(hs, hx.frame_params(mode=hx.MODE_SINGLE))
    for key, want in REFERENCE_CALLS.items():
        assert cnt[key] == want, (key, cnt[key], want)
    # gprof counted the kd builder's BBox::testIntersect calls (through BBox::intersectTriangle,
    # reference src/bbox.h:184-192) together with the render's
    lib = hx.host_lib()
    lib.hxh_scene_kd_build_box_tests.restype = ctypes.c_longlong
    lib.hxh_scene_kd_build_box_tests.argtypes = [ctypes.c_void_p, ctypes.c_int]
    build = sum(lib.hxh_scene_kd_build_box_tests(hs._h, m) for m in range(len(hs.kd_stats())))
    assert cnt["bbox_tests"] + build == REFERENCE_BBOX_TESTS, (cnt["bbox_tests"], build)
