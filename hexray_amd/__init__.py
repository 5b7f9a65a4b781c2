"""hexray_amd -- MI355X-native render path for heX-Ray scenes.

Thin ctypes binding over the two native libraries of the product:

* ``lib/libhexray_host.so`` -- the host side that stays on the CPU: ``.hexray`` loader, scene
  flattening, camera, bitmap output (``hexray_amd/host``; mirrors reference ``src/scene.cpp``,
  ``src/mesh.cpp:49-122,265-396``, ``src/camera.cpp:30-63``, ``src/bitmap.cpp``).
* ``lib/libhexray_hip.so`` -- the C-ABI render module of hand-written HIP kernels for gfx950
  (``include/hexray_hip.h``; replaces reference ``render()``, ``src/main.cpp:416-426``).

There is no CPU fallback: rendering without the HIP library or without a GPU raises.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import POINTER, c_char_p, c_double, c_float, c_int, c_int32, c_size_t, c_uint64, c_void_p

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
HOST_LIB_PATH = os.path.join(_HERE, "lib", "libhexray_host.so")
HIP_LIB_PATH = os.environ.get("HXR_HIP_LIB") or os.path.join(_HERE, "lib", "libhexray_hip.so")   # override: A/B of kernel builds

MODE_AUTO, MODE_SINGLE, MODE_ADAPTIVE_AA, MODE_MONTE_CARLO, MODE_PREPASS = 0, 1, 2, 3, 4
LAYOUT_FRAME, LAYOUT_TILES = 0, 1
DEFAULT_SEED = 5489


class FrameParams(ctypes.Structure):
    """hxr_frame_params (include/hexray_hip.h)."""

    _fields_ = [
        ("width", c_int32), ("height", c_int32), ("mode", c_int32), ("spp", c_int32),
        ("rng_seed", c_uint64),
        ("tile_size", c_int32), ("shard_rank", c_int32), ("shard_count", c_int32), ("layout", c_int32),
        ("collect_counters", c_int32), ("foveated_radius", c_int32), ("vip_x", c_int32), ("vip_y", c_int32), ("reserved_", c_int32 * 2),
    ]


COUNTER_FIELDS = [
    "camera_samples", "raycasts", "shadow_queries", "node_tests", "bbox_tests", "kd_visits", "leaf_indices",
    "tri_tests", "mesh_hits", "tex_bilinear", "env_bilinear", "hf_calls", "hf_highest", "hf_height",
    "hf_tri_tests", "hf_normals", "aa_pixels", "box_test_mismatch",
]


ACCEL_REFERENCE, ACCEL_SAH, ACCEL_REFERENCE_DEVICE_BUILD = 0, 1, 2


class SceneOptions(ctypes.Structure):
    """hxr_scene_options (include/hexray_hip.h)."""

    _fields_ = [("struct_size", ctypes.c_uint32), ("accel", ctypes.c_int32), ("no_adaptive_schedule", ctypes.c_int32), ("deferred_rays", ctypes.c_int32),
                ("device_prepare_triangles", ctypes.c_int32)]


class FrameTiming(ctypes.Structure):
    """hxr_frame_timing (include/hexray_hip.h)."""

    _fields_ = [("total_ms", c_double), ("render_ms", c_double), ("continue_ms", c_double), ("resolve_ms", c_double),
                ("deferred_rays", c_uint64), ("sched_front_items", c_uint64), ("sched_back_items", c_uint64), ("sched_part_pixels", c_uint64)]

    def as_dict(self):
        return {"total_ms": self.total_ms, "render_ms": self.render_ms, "continue_ms": self.continue_ms, "resolve_ms": self.resolve_ms,
                "deferred_rays": int(self.deferred_rays), "sched_front_items": int(self.sched_front_items), "sched_back_items": int(self.sched_back_items),
                "sched_part_pixels": int(self.sched_part_pixels)}


class Counters(ctypes.Structure):
    """hxr_counters (include/hexray_hip.h)."""

    _fields_ = [(n, c_uint64) for n in COUNTER_FIELDS] + [("reserved_", c_uint64 * 6)]

    def as_dict(self):
        return {n: int(getattr(self, n)) for n in COUNTER_FIELDS}


def frame_params(width=0, height=0, mode=MODE_AUTO, spp=0, seed=DEFAULT_SEED, tile_size=0, shard_rank=0,
                 shard_count=1, layout=LAYOUT_FRAME, collect_counters=False, foveated_radius=0, vip=(-100, -100)) -> FrameParams:
    p = FrameParams()
    p.foveated_radius, p.vip_x, p.vip_y = foveated_radius, vip[0], vip[1]
    p.width, p.height, p.mode, p.spp = width, height, mode, spp
    p.rng_seed = seed
    p.tile_size, p.shard_rank, p.shard_count, p.layout = tile_size, shard_rank, shard_count, layout
    p.collect_counters = 1 if collect_counters else 0
    return p


_host = None


def host_lib():
    """libhexray_host.so (loader + flattener); raises if it has not been built."""
    global _host
    if _host is None:
        if not os.path.exists(HOST_LIB_PATH):
            raise RuntimeError(f"{HOST_LIB_PATH} is missing: run ./build.sh (or __graft_entry__.build())")
        lib = ctypes.CDLL(HOST_LIB_PATH)
        lib.hxh_scene_load.restype = c_void_p
        lib.hxh_scene_load.argtypes = [c_char_p, c_char_p, c_char_p, c_size_t]
        lib.hxh_scene_load_ex.restype = c_void_p
        lib.hxh_scene_load_ex.argtypes = [c_char_p, c_char_p, ctypes.c_uint, c_char_p, c_size_t]
        lib.hxh_scene_desc.restype = c_void_p
        lib.hxh_scene_desc.argtypes = [c_void_p]
        lib.hxh_scene_free.argtypes = [c_void_p]
        for name in ("hxh_scene_num_threads", "hxh_scene_prepass_samples", "hxh_scene_interactive", "hxh_scene_num_warnings"):
            getattr(lib, name).argtypes = [c_void_p]
        lib.hxh_scene_warning.restype = c_char_p
        lib.hxh_scene_warning.argtypes = [c_void_p, c_int]
        lib.hxh_scene_kd_stats.argtypes = [c_void_p, c_int, POINTER(c_int), POINTER(c_int), POINTER(c_int), POINTER(c_double), POINTER(c_double)]
        lib.hxh_camera_move.argtypes = [c_void_p, c_double, c_double]
        lib.hxh_camera_rotate.argtypes = [c_void_p, c_double, c_double]
        lib.hxh_bucket_list.argtypes = [c_int, c_int, c_int, POINTER(c_int), c_int]
        lib.hxh_save_bmp.argtypes = [c_char_p, c_void_p, c_int, c_int, c_size_t]
        lib.hxh_save_raw.argtypes = [c_char_p, c_void_p, c_int, c_int, c_size_t]
        lib.hxh_save_exr.argtypes = [c_char_p, c_void_p, c_int, c_int, c_size_t]
        lib.hxh_float_to_half.restype = ctypes.c_uint
        lib.hxh_float_to_half.argtypes = [c_float]
        lib.hxh_srgb8.restype = ctypes.c_uint
        lib.hxh_srgb8.argtypes = [c_float]
        lib.hxh_load_image.restype = POINTER(c_float)
        lib.hxh_load_image.argtypes = [c_char_p, POINTER(c_int), POINTER(c_int)]
        lib.hxh_free_pixels.argtypes = [POINTER(c_float)]
        _host = lib
    return _host


class HostScene:
    """A parsed, prepared and flattened ``.hexray`` scene (host memory).

    ``overlay`` is extra scene text in the same syntax whose blocks override properties of the
    matching blocks of the file -- the reference can only take W x H / spp / AA from the scene file
    (reference src/scene.cpp:764-778, src/camera.h:57-72), so overrides are expressed the same way.
    """

    def __init__(self, path: str, overlay: str = "", gpu_heightfield_tables: bool = False, gpu_kd_build: bool = False,
                 gpu_prepare_triangles: bool = False, gpu_heightfield_blur: bool = False, gpu_bitmap_prep: bool = False):
        """``gpu_heightfield_tables``: leave the heightfield maxima tables to the render module (it builds them on the GPU); such a
        scene is for the HIP module only -- the CPU oracle needs the host's tables.
        ``gpu_kd_build``: skip Mesh::buildKD on the host; create the DeviceScene with ACCEL_REFERENCE_DEVICE_BUILD (or ACCEL_SAH).
        ``gpu_prepare_triangles``: skip Mesh::prepareTriangles; create the DeviceScene with ``device_prepare_triangles=True``.
        ``gpu_heightfield_blur``: leave a heightfield's blur, bounding box, normals and maxima tables to the render module.
        ``gpu_bitmap_prep``: leave Bitmap::decompressGamma / Bitmap::differentiate to the render module (hxr_bitmap.prep)."""
        lib = host_lib()
        err = ctypes.create_string_buffer(1024)
        self._h = lib.hxh_scene_load_ex(os.fsencode(path), overlay.encode(), (1 if gpu_heightfield_tables else 0) | (2 if gpu_kd_build else 0) | (4 if gpu_prepare_triangles else 0)
                                        | (8 if gpu_heightfield_blur else 0) | (16 if gpu_bitmap_prep else 0),
                                        err, len(err))
        if not self._h:
            raise ValueError(f"cannot load scene {path}: {err.value.decode(errors='replace')}")
        self.path = path

    @property
    def desc(self) -> int:
        """Address of the hxr_scene_desc (valid while this object lives)."""
        return host_lib().hxh_scene_desc(self._h)

    @property
    def settings(self):
        return SceneDescHeader.from_address(self.desc).settings

    @property
    def camera(self):
        return SceneDescHeader.from_address(self.desc).camera

    @property
    def warnings(self):
        lib = host_lib()
        return [lib.hxh_scene_warning(self._h, i).decode() for i in range(lib.hxh_scene_num_warnings(self._h))]

    def kd_stats(self):
        """Per mesh: (nodes, leaf nodes, max depth, avg depth, avg tris per leaf) as Mesh::beginRender prints them."""
        lib, out, m = host_lib(), [], 0
        while True:
            a, b, c, d, e = c_int(), c_int(), c_int(), c_double(), c_double()
            if lib.hxh_scene_kd_stats(self._h, m, a, b, c, d, e) != 0:
                return out
            out.append((a.value, b.value, c.value, d.value, e.value))
            m += 1

    def move_camera(self, sideways: float, front_back: float):
        host_lib().hxh_camera_move(self._h, sideways, front_back)

    def rotate_camera(self, yaw: float, pitch: float):
        host_lib().hxh_camera_rotate(self._h, yaw, pitch)

    def close(self):
        if getattr(self, "_h", None):
            host_lib().hxh_scene_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Settings(ctypes.Structure):
    _fields_ = [("ambient", c_float * 3), ("background", c_float * 3), ("max_trace_depth", c_int32), ("want_aa", c_int32),
                ("gi", c_int32), ("num_paths", c_int32), ("frame_width", c_int32), ("frame_height", c_int32)]


class Camera(ctypes.Structure):
    _fields_ = [("pos", c_double * 3), ("top_left", c_double * 3), ("top_right", c_double * 3), ("bottom_left", c_double * 3),
                ("up", c_double * 3), ("right", c_double * 3), ("front", c_double * 3), ("aperture_size", c_double),
                ("focal_plane_dist", c_double), ("stereo_separation", c_double), ("dof", c_int32), ("num_samples", c_int32),
                ("auto_focus", c_int32), ("pad_", c_int32)]


class SceneDescHeader(ctypes.Structure):
    """Leading fields of hxr_scene_desc (enough to read settings / camera from Python)."""

    _fields_ = [("abi_version", ctypes.c_uint32), ("reserved_", ctypes.c_uint32), ("settings", Settings), ("camera", Camera)]


def bucket_list(width: int, height: int, bucket_size: int = 64):
    """The reference's zig-zag bucket order (src/sdl.cpp:280-294) as an (n, 4) int array x0,y0,x1,y1."""
    lib = host_lib()
    n = lib.hxh_bucket_list(width, height, bucket_size, None, 0)
    buf = (c_int * (4 * n))()
    lib.hxh_bucket_list(width, height, bucket_size, buf, n)
    return np.frombuffer(buf, dtype=np.int32).reshape(n, 4).copy()


def save_bmp(path: str, frame: np.ndarray):
    """Write a float RGB frame as 24-bit sRGB BMP exactly like Bitmap::saveBMP (src/bitmap.cpp:190-240)."""
    f = np.ascontiguousarray(frame, dtype=np.float32)
    h, w, _ = f.shape
    if host_lib().hxh_save_bmp(os.fsencode(path), f.ctypes.data, w, h, w * 12) != 0:
        raise OSError(f"cannot write {path}")


def save_exr(path: str, frame: np.ndarray):
    """Write a float RGB frame as an OpenEXR file like Bitmap::saveEXR (src/bitmap.cpp:270-288): RGBA HALF, alpha 1."""
    f = np.ascontiguousarray(frame, dtype=np.float32)
    h, w, _ = f.shape
    if host_lib().hxh_save_exr(os.fsencode(path), f.ctypes.data, w, h, w * 12) != 0:
        raise OSError(f"cannot write {path}")


def load_image(path: str) -> np.ndarray:
    lib = host_lib()
    w, h = c_int(), c_int()
    p = lib.hxh_load_image(os.fsencode(path), w, h)
    if not p:
        raise OSError(f"cannot load {path}")
    try:
        return np.ctypeslib.as_array(p, shape=(h.value, w.value, 3)).copy()
    finally:
        lib.hxh_free_pixels(p)


_hip = None


def hip_lib():
    """libhexray_hip.so, the C-ABI render module.  Raises if it is missing -- there is no fallback."""
    global _hip
    if _hip is None:
        if not os.path.exists(HIP_LIB_PATH):
            raise RuntimeError(f"{HIP_LIB_PATH} is missing: run ./build.sh (or __graft_entry__.build()); there is no CPU fallback")
        # When PyTorch shares the process (bench.py, tests) its bundled HIP runtime must be the one that gets
        # loaded: two copies of libamdhip64 in one process leave the second without devices.
        try:
            import torch  # noqa: F401
        except Exception:
            pass
        lib = ctypes.CDLL(HIP_LIB_PATH)
        lib.hxr_device_count.restype = c_int
        lib.hxr_init.argtypes = [c_int]
        lib.hxr_scene_create.restype = c_void_p
        lib.hxr_scene_create.argtypes = [c_void_p]
        lib.hxr_scene_create_ex.restype = c_void_p
        lib.hxr_scene_create_ex.argtypes = [c_void_p, POINTER(SceneOptions)]
        lib.hxr_scene_tree_stats.argtypes = [c_void_p, c_int, POINTER(ctypes.c_int64), POINTER(ctypes.c_int64), POINTER(c_int)]
        lib.hxr_kd_build_device.argtypes = [c_void_p, c_int, c_void_p, ctypes.c_int64, c_void_p, ctypes.c_int64, POINTER(ctypes.c_int64), POINTER(ctypes.c_int64),
                                            POINTER(c_int), POINTER(c_double)]
        lib.hxr_sah_build_device.argtypes = [c_void_p, c_int, c_void_p, ctypes.c_int64, c_void_p, ctypes.c_int64, POINTER(ctypes.c_int64), POINTER(ctypes.c_int64),
                                            POINTER(c_int), POINTER(c_double)]
        lib.hxr_sah_build.argtypes = [c_void_p, c_int, c_void_p, ctypes.c_int64, c_void_p, ctypes.c_int64, POINTER(ctypes.c_int64), POINTER(ctypes.c_int64), POINTER(c_int)]
        lib.hxr_scene_update_camera.argtypes = [c_void_p, c_void_p]
        lib.hxr_scene_verify_tables.argtypes = [c_void_p, c_void_p, POINTER(ctypes.c_int64)]
        lib.hxr_render.argtypes = [c_void_p, POINTER(FrameParams), c_void_p, c_size_t, c_void_p, c_void_p, c_void_p]
        lib.hxr_render_device.argtypes = [c_void_p, POINTER(FrameParams), c_void_p, c_size_t, c_void_p]
        lib.hxr_output_bytes.restype = c_size_t
        lib.hxr_output_bytes.argtypes = [c_void_p, POINTER(FrameParams), c_size_t]
        lib.hxr_shard_tile_count.argtypes = [c_void_p, POINTER(FrameParams)]
        lib.hxr_shard_tile_rect.argtypes = [c_void_p, POINTER(FrameParams), c_int] + [POINTER(c_int)] * 4
        lib.hxr_assemble_frame.argtypes = [c_void_p, POINTER(FrameParams), c_void_p, c_size_t, c_void_p, c_size_t, c_void_p]
        lib.hxr_quantize_srgb8.argtypes = [c_void_p, c_size_t, c_int, c_int, c_void_p, c_void_p]
        lib.hxr_get_counters.argtypes = [c_void_p, POINTER(Counters)]
        lib.hxr_last_kernel_ms.restype = c_double
        lib.hxr_last_kernel_ms.argtypes = [c_void_p]
        lib.hxr_get_frame_timing.argtypes = [c_void_p, POINTER(FrameTiming)]
        lib.hxr_interactive_frame.argtypes = [c_void_p, POINTER(FrameParams), c_void_p, c_void_p, c_size_t, c_int, c_int, c_int, POINTER(c_double)]
        lib.hxr_interactive_float_frame.argtypes = [c_void_p, POINTER(FrameParams), c_void_p, c_size_t]
        lib.hxr_group_create.restype = c_void_p
        lib.hxr_group_create.argtypes = [c_void_p, POINTER(SceneOptions), POINTER(c_int), c_int]
        lib.hxr_group_update_camera.argtypes = [c_void_p, c_void_p]
        lib.hxr_group_render.argtypes = [c_void_p, POINTER(FrameParams), c_void_p, c_size_t, c_void_p, c_void_p, c_void_p]
        lib.hxr_group_last_ms.restype = c_double
        lib.hxr_group_last_ms.argtypes = [c_void_p, POINTER(c_double)]
        lib.hxr_group_size.argtypes = [c_void_p]
        lib.hxr_group_destroy.argtypes = [c_void_p]
        lib.hxr_scene_destroy.argtypes = [c_void_p]
        lib.hxr_last_error.restype = c_char_p
        _hip = lib
    return _hip


class HipError(RuntimeError):
    pass


class Cancelled(RuntimeError):
    """hxr_render / hxr_group_render returned HXR_CANCELLED: the caller's flag was raised, the frame is incomplete."""


PROGRESS_CB = ctypes.CFUNCTYPE(None, c_void_p, c_int, c_int, c_int, c_int)   # hxr_progress_cb


def _check(code: int, what: str):
    if code < 0:
        raise HipError(f"{what} failed ({code}): {hip_lib().hxr_last_error().decode(errors='replace')}")
    return code


_initialised_device = None


def init(device_id: int = 0):
    """Bind this process to one GPU (one process per GPU)."""
    global _initialised_device
    lib = hip_lib()
    if lib.hxr_device_count() <= 0:
        raise HipError("no gfx950 device visible: the render path has no CPU fallback")
    _check(lib.hxr_init(device_id), "hxr_init")
    _initialised_device = device_id


KD_NODE_DTYPE = np.dtype([("axis", "<i4"), ("first", "<i4"), ("count", "<i4"), ("parent", "<i4"), ("split", "<f8")])   # hxr_kd_node


def kd_build_device(host_scene: "HostScene", mesh: int = 0):
    """Mesh::buildKD on the GPU (hxr_kd_build_device): the reference's tree of mesh ``mesh`` of the scene, built from its triangles.
    Returns (nodes [KD_NODE_DTYPE], leaf_indices [int32], max_depth, build_ms)."""
    lib = hip_lib()
    nn, nl, depth, ms = ctypes.c_int64(), ctypes.c_int64(), c_int(), c_double()
    _check(lib.hxr_kd_build_device(host_scene.desc, mesh, None, 0, None, 0, nn, nl, depth, ms), "hxr_kd_build_device")
    nodes = np.zeros(nn.value, dtype=KD_NODE_DTYPE)
    leaf = np.zeros(max(1, nl.value), dtype=np.int32)
    _check(lib.hxr_kd_build_device(host_scene.desc, mesh, nodes.ctypes.data, nn.value, leaf.ctypes.data, leaf.size, nn, nl, depth, ms), "hxr_kd_build_device")
    return nodes, leaf[:nl.value], depth.value, ms.value


def sah_build(host_scene: "HostScene", mesh: int = 0, device: bool = True):
    """The module's SAH kd-tree of mesh ``mesh``: built by kernels (hxr_sah_build_device) or by the host builder (hxr_sah_build).
    Returns (nodes [KD_NODE_DTYPE], leaf_indices [int32], max_depth, build_ms or None)."""
    lib = hip_lib()
    nn, nl, depth, ms = ctypes.c_int64(), ctypes.c_int64(), c_int(), c_double()
    if device:
        _check(lib.hxr_sah_build_device(host_scene.desc, mesh, None, 0, None, 0, nn, nl, depth, ms), "hxr_sah_build_device")
    else:
        _check(lib.hxr_sah_build(host_scene.desc, mesh, None, 0, None, 0, nn, nl, depth), "hxr_sah_build")
    nodes = np.zeros(nn.value, dtype=KD_NODE_DTYPE)
    leaf = np.zeros(max(1, nl.value), dtype=np.int32)
    if device:
        _check(lib.hxr_sah_build_device(host_scene.desc, mesh, nodes.ctypes.data, nn.value, leaf.ctypes.data, leaf.size, nn, nl, depth, ms), "hxr_sah_build_device")
    else:
        _check(lib.hxr_sah_build(host_scene.desc, mesh, nodes.ctypes.data, nn.value, leaf.ctypes.data, leaf.size, nn, nl, depth), "hxr_sah_build")
    return nodes, leaf[:nl.value], depth.value, (ms.value if device else None)


class DeviceScene:
    """A scene resident in HBM (hxr_scene).  ``accel``: ACCEL_REFERENCE walks the kd-trees of the descriptor (the
    reference's own), ACCEL_SAH rebuilds them with the surface-area heuristic (hxr_scene_create_ex)."""

    def __init__(self, host_scene: HostScene, accel: int = 0, adaptive_schedule: bool = True, deferred_rays: bool = False,
                 device_prepare_triangles: bool = False):
        """``device_prepare_triangles``: Mesh::prepareTriangles runs on the GPU (the descriptor's derived triangle fields are not read)."""
        if _initialised_device is None:
            init(0)
        if accel == ACCEL_REFERENCE and adaptive_schedule and not deferred_rays and not device_prepare_triangles:
            self._s = hip_lib().hxr_scene_create(host_scene.desc)
        else:
            opt = SceneOptions(ctypes.sizeof(SceneOptions), accel, 0 if adaptive_schedule else 1, 1 if deferred_rays else 0, 1 if device_prepare_triangles else 0)
            self._s = hip_lib().hxr_scene_create_ex(host_scene.desc, ctypes.byref(opt))
        if not self._s:
            raise HipError("hxr_scene_create failed: " + hip_lib().hxr_last_error().decode(errors="replace"))
        self.host = host_scene

    def tree_stats(self, mesh: int):
        """(nodes, leaf entries, max depth) of the tree walked for mesh ``mesh``."""
        n, l, dpt = ctypes.c_int64(), ctypes.c_int64(), c_int()
        _check(hip_lib().hxr_scene_tree_stats(self._s, mesh, ctypes.byref(n), ctypes.byref(l), ctypes.byref(dpt)), "hxr_scene_tree_stats")
        return n.value, l.value, dpt.value

    def update_camera(self):
        cam_addr = ctypes.addressof(SceneDescHeader.from_address(self.host.desc).camera)
        _check(hip_lib().hxr_scene_update_camera(self._s, cam_addr), "hxr_scene_update_camera")

    def verify_tables(self, complete_host_scene: HostScene | None = None) -> int:
        """Number of GPU-prepared table entries (hot triangle records, heightfield maxima) that differ from the host's."""
        n = ctypes.c_int64()
        _check(hip_lib().hxr_scene_verify_tables(self._s, (complete_host_scene or self.host).desc, ctypes.byref(n)), "hxr_scene_verify_tables")
        return n.value

    def frame_size(self, params: FrameParams):
        s = self.host.settings
        return (params.width or s.frame_width, params.height or s.frame_height)

    def render(self, params: FrameParams | None = None, cancel=None, on_tile=None, out: np.ndarray | None = None) -> np.ndarray:
        """Blocking render into a host float32 array of shape (H, W, 3): the reference's vfb.  ``cancel``: a ctypes.c_int the
        caller may raise from another thread (the call then raises Cancelled); ``on_tile(x0, y0, x1, y1)``: progress callback."""
        params = params or frame_params()
        w, h = self.frame_size(params)
        if out is None:
            out = np.zeros((h, w, 3), dtype=np.float32)
        cb = PROGRESS_CB(lambda user, x0, y0, x1, y1: on_tile(x0, y0, x1, y1)) if on_tile else None
        rc = _check(hip_lib().hxr_render(self._s, ctypes.byref(params), out.ctypes.data, w * 12, ctypes.byref(cancel) if cancel is not None else None, cb, None), "hxr_render")
        if rc == 1:
            raise Cancelled("hxr_render: cancelled")
        return out

    def interactive_frame(self, params: FrameParams | None = None, update_camera: bool = True, shifts=(16, 8, 0)):
        """One frame of the interactive loop: camera update, render, conversion to a 32-bit display surface on the GPU.
        Returns (surface uint32 [H, W], latency in ms)."""
        params = params or frame_params()
        w, h = self.frame_size(params)
        surface = np.zeros((h, w), dtype=np.uint32)
        cam_addr = ctypes.addressof(SceneDescHeader.from_address(self.host.desc).camera) if update_camera else None
        ms = c_double()
        _check(hip_lib().hxr_interactive_frame(self._s, ctypes.byref(params), cam_addr, surface.ctypes.data, w * 4, shifts[0], shifts[1], shifts[2], ctypes.byref(ms)),
               "hxr_interactive_frame")
        return surface, ms.value

    def interactive_float_frame(self, params: FrameParams | None = None) -> np.ndarray:
        params = params or frame_params()
        w, h = self.frame_size(params)
        out = np.zeros((h, w, 3), dtype=np.float32)
        _check(hip_lib().hxr_interactive_float_frame(self._s, ctypes.byref(params), out.ctypes.data, w * 12), "hxr_interactive_float_frame")
        return out

    def render_device(self, params: FrameParams, d_out: int, row_stride_bytes: int, stream: int = 0):
        """Asynchronous render into device memory at address ``d_out`` on HIP stream ``stream``."""
        _check(hip_lib().hxr_render_device(self._s, ctypes.byref(params), d_out, row_stride_bytes, stream), "hxr_render_device")

    def output_bytes(self, params: FrameParams, row_stride_bytes: int) -> int:
        return hip_lib().hxr_output_bytes(self._s, ctypes.byref(params), row_stride_bytes)

    def shard_tiles(self, params: FrameParams):
        lib = hip_lib()
        n = lib.hxr_shard_tile_count(self._s, ctypes.byref(params))
        rects = np.zeros((n, 4), dtype=np.int32)
        for i in range(n):
            a, b, c, d = c_int(), c_int(), c_int(), c_int()
            lib.hxr_shard_tile_rect(self._s, ctypes.byref(params), i, a, b, c, d)
            rects[i] = (a.value, b.value, c.value, d.value)
        return rects

    def assemble_frame(self, params: FrameParams, d_gathered: int, per_rank_bytes: int, d_frame: int, row_stride_bytes: int, stream: int = 0):
        _check(hip_lib().hxr_assemble_frame(self._s, ctypes.byref(params), d_gathered, per_rank_bytes, d_frame, row_stride_bytes, stream), "hxr_assemble_frame")

    def counters(self) -> dict:
        c = Counters()
        _check(hip_lib().hxr_get_counters(self._s, ctypes.byref(c)), "hxr_get_counters")
        return c.as_dict()

    def counters_raw(self) -> Counters:
        """The whole hxr_counters record (reserved_ holds the phase clocks of a -DHXR_PHASE_TIMERS build)."""
        c = Counters()
        _check(hip_lib().hxr_get_counters(self._s, ctypes.byref(c)), "hxr_get_counters")
        return c

    def last_kernel_ms(self) -> float:
        return hip_lib().hxr_last_kernel_ms(self._s)

    def frame_timing(self) -> dict:
        """Where the last frame's GPU time went: render / continuation / resolve kernels, deferred ray count."""
        t = FrameTiming()
        _check(hip_lib().hxr_get_frame_timing(self._s, ctypes.byref(t)), "hxr_get_frame_timing")
        return t.as_dict()

    def close(self):
        if getattr(self, "_s", None):
            hip_lib().hxr_scene_destroy(self._s)
            self._s = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class DeviceGroup:
    """One process, several GPUs (hxr_group_*): the frame's tiles are dealt round-robin to the devices, gathered with
    peer-to-peer copies and assembled on the first one.  A device id may be listed more than once."""

    def __init__(self, host_scene: HostScene, devices, accel: int = 0, adaptive_schedule: bool = True, deferred_rays: bool = False):
        opt = SceneOptions(ctypes.sizeof(SceneOptions), accel, 0 if adaptive_schedule else 1, 1 if deferred_rays else 0)
        ids = (c_int * len(devices))(*devices)
        self._g = hip_lib().hxr_group_create(host_scene.desc, ctypes.byref(opt), ids, len(devices))
        if not self._g:
            raise HipError("hxr_group_create failed: " + hip_lib().hxr_last_error().decode(errors="replace"))
        self.host = host_scene
        # hxr_group_create leaves the process-wide device selection as it found it: a DeviceScene created later still calls init()

    def render(self, params: FrameParams | None = None, cancel=None, on_tile=None) -> np.ndarray:
        params = params or frame_params()
        s = self.host.settings
        w, h = params.width or s.frame_width, params.height or s.frame_height
        out = np.zeros((h, w, 3), dtype=np.float32)
        cb = PROGRESS_CB(lambda user, x0, y0, x1, y1: on_tile(x0, y0, x1, y1)) if on_tile else None
        rc = _check(hip_lib().hxr_group_render(self._g, ctypes.byref(params), out.ctypes.data, w * 12, ctypes.byref(cancel) if cancel is not None else None, cb, None),
                    "hxr_group_render")
        if rc == 1:
            raise Cancelled("hxr_group_render: cancelled")
        return out

    def last_ms(self):
        n = hip_lib().hxr_group_size(self._g)
        each = (c_double * n)()
        total = hip_lib().hxr_group_last_ms(self._g, each)
        return total, list(each)

    def update_camera(self):
        cam_addr = ctypes.addressof(SceneDescHeader.from_address(self.host.desc).camera)
        _check(hip_lib().hxr_group_update_camera(self._g, cam_addr), "hxr_group_update_camera")

    def close(self):
        if getattr(self, "_g", None):
            hip_lib().hxr_group_destroy(self._g)
            self._g = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def overlay_text(width=None, height=None, want_aa=None, dof=None, num_samples=None, gi=None, num_paths=None,
                 prepass_samples=0, interactive=False, extra: str = "") -> str:
    """Build the overlay text for the configs of SURVEY.md section 8d."""
    g, c = [], []
    if width is not None:
        g.append(f"frameWidth {width}")
    if height is not None:
        g.append(f"frameHeight {height}")
    if want_aa is not None:
        g.append(f"wantAA {'on' if want_aa else 'off'}")
    if gi is not None:
        g.append(f"gi {'on' if gi else 'off'}")
    if num_paths is not None:
        g.append(f"numPaths {num_paths}")
    if prepass_samples is not None:
        g.append(f"prepassSamples {prepass_samples}")
    if interactive is not None:
        g.append(f"interactive {'on' if interactive else 'off'}")
    if dof is not None:
        c.append(f"dof {'on' if dof else 'off'}")
    if num_samples is not None:
        c.append(f"numSamples {num_samples}")
    text = ""
    if g:
        text += "GlobalSettings {\n" + "\n".join(g) + "\n}\n"
    if c:
        text += "Camera {\n" + "\n".join(c) + "\n}\n"
    return text + extra
