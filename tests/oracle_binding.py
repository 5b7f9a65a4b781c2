"""ctypes binding of the CPU oracle (oracle/liboracle.so).  TEST INFRASTRUCTURE: imported only by
tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg -- never by hexray_amd/."""
import ctypes
import os
from ctypes import POINTER, c_double, c_float, c_int, c_size_t, c_uint32, c_uint64, c_void_p

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_LIB_PATH = os.path.join(ROOT, "oracle", "liboracle.so")

_lib = None


class ProbeHit(ctypes.Structure):
    """hxo_probe_hit (oracle/hxo_oracle.c): the fields of IntersectionInfo a shader or texture reads."""

    _fields_ = [("ip", c_double * 3), ("norm", c_double * 3), ("dNdx", c_double * 3), ("dNdy", c_double * 3), ("u", c_double), ("v", c_double)]


def oracle_lib(path=None):
    global _lib
    if path is not None:
        return _load(path)
    if _lib is None:
        _lib = _load(ORACLE_LIB_PATH)
    return _lib


def _load(path):
    from hexray_amd import Counters, FrameParams
    if not os.path.exists(path):
        raise RuntimeError(f"{path} is missing: run ./build.sh oracle")
    lib = ctypes.CDLL(path)
    lib.hxo_render.argtypes = [c_void_p, POINTER(FrameParams), c_void_p, c_size_t, c_int, POINTER(Counters)]
    lib.hxo_geometry_intersect.argtypes = [c_void_p, c_int, POINTER(c_double), POINTER(c_double), POINTER(c_double)]
    lib.hxo_node_intersect.argtypes = [c_void_p, c_int, POINTER(c_double), POINTER(c_double), POINTER(c_double)]
    lib.hxo_box_test.argtypes = [POINTER(c_double)] * 4 + [POINTER(c_double)]
    lib.hxo_light_intersect.argtypes = [c_void_p, c_int, POINTER(c_double), POINTER(c_double), POINTER(c_double)]
    lib.hxo_light_solid_angle.restype = c_double
    lib.hxo_light_solid_angle.argtypes = [c_void_p, c_int, POINTER(c_double)]
    lib.hxo_light_sample.argtypes = [c_void_p, c_int, c_int, POINTER(c_double), c_double, c_double, POINTER(c_double), POINTER(c_float)]
    lib.hxo_camera_ray.argtypes = [c_void_p, c_int, c_int, c_double, c_double, c_int, c_double, c_double, POINTER(c_double)]
    lib.hxo_philox_words.argtypes = [c_uint64, c_uint32, c_uint32, c_uint32, c_uint32, POINTER(c_uint32)]
    lib.hxo_max_threads.restype = c_int
    # unit-level probes of the shading members (tests/test_golden_shading.py)
    lib.hxo_box_split.argtypes = [POINTER(c_double), POINTER(c_double), c_int, c_double, POINTER(c_double)]
    lib.hxo_hemisphere.argtypes = [c_double, c_double, POINTER(c_double), POINTER(c_double)]
    lib.hxo_disk_map.argtypes = [c_double, c_double, POINTER(c_double)]
    lib.hxo_glossy_dir.argtypes = [POINTER(c_double), POINTER(c_double), c_float, c_double, c_double, POINTER(c_double)]
    lib.hxo_texture_sample.argtypes = [c_void_p, c_int, POINTER(c_double), POINTER(ProbeHit), POINTER(c_float)]
    lib.hxo_texture_bump.argtypes = [c_void_p, c_int, POINTER(ProbeHit)]
    lib.hxo_shade_probe.argtypes = [c_void_p, c_int, POINTER(c_double), POINTER(c_double), c_int, c_uint32, POINTER(ProbeHit), POINTER(c_float), POINTER(c_double), c_int]
    lib.hxo_brdf_eval.argtypes = [c_void_p, c_int, POINTER(ProbeHit), POINTER(c_double), POINTER(c_double), POINTER(c_float)]
    lib.hxo_brdf_spawn.argtypes = [c_void_p, c_int, POINTER(ProbeHit), POINTER(c_double), POINTER(c_double), c_int, c_uint32, POINTER(c_double), POINTER(c_float), POINTER(c_float)]
    lib.hxo_shade_probe2.argtypes = [c_void_p, c_int, POINTER(c_double), POINTER(c_double), c_int, c_uint32, POINTER(ProbeHit), POINTER(c_float), POINTER(c_double), c_int,
                                     c_uint64, POINTER(c_double), c_int, POINTER(c_int), POINTER(c_double), c_int, POINTER(c_int)]
    lib.hxo_brdf_spawn2.argtypes = [c_void_p, c_int, POINTER(ProbeHit), POINTER(c_double), POINTER(c_double), c_int, c_uint32, c_double, c_double, POINTER(c_double),
                                    POINTER(c_float), POINTER(c_float)]
    return lib


def oracle_render(host_scene, params=None, threads=0, lib=None):
    """Render with the CPU oracle.  Returns (frame float32 [H,W,3], counters dict)."""
    from hexray_amd import Counters, frame_params
    lib = lib or oracle_lib()
    params = params or frame_params()
    s = host_scene.settings
    w, h = params.width or s.frame_width, params.height or s.frame_height
    out = np.zeros((h, w, 3), dtype=np.float32)
    cnt = Counters()
    rc = lib.hxo_render(host_scene.desc, ctypes.byref(params), out.ctypes.data, w * 12, threads, ctypes.byref(cnt))
    if rc != 0:
        raise RuntimeError(f"hxo_render failed ({rc})")
    return out, cnt.as_dict()


def dvec(*v):
    return (c_double * len(v))(*v)
