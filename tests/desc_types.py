"""ctypes mirror of the record types of include/hexray_hip.h (tests only: lets a test read the flattened
scene the host library produced)."""
import ctypes
from ctypes import POINTER, c_double, c_float, c_int32, c_int64, c_uint32

from hexray_amd import Camera, Settings


class Transform(ctypes.Structure):
    _fields_ = [("offset", c_double * 3), ("m", c_double * 9), ("inv_m", c_double * 9), ("tinv_m", c_double * 9)]


class Geometry(ctypes.Structure):
    _fields_ = [("type", c_int32), ("left", c_int32), ("right", c_int32), ("ref", c_int32), ("p", c_double * 6)]


class Triangle(ctypes.Structure):
    _fields_ = [("v", c_int32 * 3), ("n", c_int32 * 3), ("t", c_int32 * 3), ("pad_", c_int32), ("gnormal", c_double * 3),
                ("AB", c_double * 3), ("AC", c_double * 3), ("ABcrossAC", c_double * 3), ("dNdx", c_double * 3), ("dNdy", c_double * 3)]


class KdNode(ctypes.Structure):
    _fields_ = [("axis", c_int32), ("first", c_int32), ("count", c_int32), ("parent", c_int32), ("split", c_double)]


class Mesh(ctypes.Structure):
    _fields_ = [("faceted", c_int32), ("backface_culling", c_int32), ("has_kd", c_int32), ("pad_", c_int32),
                ("bbox_min", c_double * 3), ("bbox_max", c_double * 3),
                ("first_vertex", c_int64), ("num_vertices", c_int64), ("first_normal", c_int64), ("num_normals", c_int64),
                ("first_uv", c_int64), ("num_uvs", c_int64), ("first_triangle", c_int64), ("num_triangles", c_int64),
                ("first_kd_node", c_int64), ("num_kd_nodes", c_int64), ("first_leaf_index", c_int64), ("num_leaf_indices", c_int64)]


class Heightfield(ctypes.Structure):
    _fields_ = [("W", c_int32), ("H", c_int32), ("max_k", c_int32), ("use_optimization", c_int32), ("bbox_min", c_double * 3),
                ("bbox_max", c_double * 3), ("first_height", c_int64), ("first_maxh", c_int64), ("first_highmap", c_int64), ("first_normal", c_int64),
                ("raw_heights", c_int32), ("blur_radius", c_int32), ("first_blur_kernel", c_int64)]


class Bitmap(ctypes.Structure):
    _fields_ = [("width", c_int32), ("height", c_int32), ("first_texel", c_int64), ("prep", c_int32), ("gamma", ctypes.c_float), ("first_gamma_table", c_int64)]


class Texture(ctypes.Structure):
    _fields_ = [("type", c_int32), ("bitmap", c_int32), ("color1", c_float * 3), ("color2", c_float * 3), ("scaling", c_double),
                ("strength", c_double), ("ior", c_double)]


class Shader(ctypes.Structure):
    _fields_ = [("type", c_int32), ("texture", c_int32), ("color", c_float * 3), ("specular", c_float * 3), ("exponent", c_float),
                ("glossiness", c_float), ("num_samples", c_int32), ("first_layer", c_int32), ("num_layers", c_int32), ("pad_", c_int32), ("ior", c_double)]


class Layer(ctypes.Structure):
    _fields_ = [("shader", c_int32), ("texture", c_int32), ("blend", c_float * 3)]


class Light(ctypes.Structure):
    _fields_ = [("type", c_int32), ("x_subd", c_int32), ("y_subd", c_int32), ("color", c_float * 3), ("power", c_float), ("scale_factor", c_float),
                ("pos", c_double * 3), ("area", c_double), ("T", Transform)]


class Node(ctypes.Structure):
    _fields_ = [("geometry", c_int32), ("shader", c_int32), ("bump", c_int32), ("pad_", c_int32), ("T", Transform)]


class Environment(ctypes.Structure):
    _fields_ = [("present", c_int32), ("loaded", c_int32), ("side_bitmap", c_int32 * 6)]


def _table(name, typ):
    return [(name, POINTER(typ)), ("num_" + name, c_int64)]


class SceneDesc(ctypes.Structure):
    _fields_ = ([("abi_version", c_uint32), ("reserved_", c_uint32), ("settings", Settings), ("camera", Camera), ("environment", Environment)]
                + _table("geometries", Geometry) + _table("meshes", Mesh) + _table("heightfields", Heightfield) + _table("triangles", Triangle)
                + _table("kd_nodes", KdNode) + _table("leaf_indices", c_int32) + _table("bitmaps", Bitmap) + _table("textures", Texture)
                + _table("shaders", Shader) + _table("layers", Layer) + _table("lights", Light) + _table("nodes", Node)
                + [("f64_pool", POINTER(c_double)), ("f64_pool_size", c_int64), ("f32_pool", POINTER(c_float)), ("f32_pool_size", c_int64)])


def desc_of(host_scene) -> SceneDesc:
    return SceneDesc.from_address(host_scene.desc)
