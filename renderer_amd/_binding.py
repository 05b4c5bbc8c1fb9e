"""ctypes view of include/rdr.h.

`CApi(lib, prefix)` binds every entry point of the header under a given symbol
prefix, so the same Python code can drive the HIP library (`rdr_*`) and, in
tests, any other library that exports the same ABI under another prefix.
"""
import ctypes as C

import numpy as np

RDR_OK = 0
RDR_ERR_INVALID, RDR_ERR_DEVICE, RDR_ERR_CAPACITY, RDR_ERR_UNSUPPORTED = 1, 2, 3, 4
INVALID_ID = 0xFFFFFFFF

# enum rdr_image_format (Renderer/Image.h:5-14)
FORMAT_R8G8B8_UNORM, FORMAT_R8G8B8_SRGB, FORMAT_R8G8B8A8_UNORM, FORMAT_R8G8B8A8_SRGB = 0, 1, 2, 3
FORMAT_R16G16B16_SFLOAT, FORMAT_R16G16B16A16_SFLOAT, FORMAT_R32G32B32_SFLOAT, FORMAT_R32G32B32A32_SFLOAT = 4, 5, 6, 7
# enum rdr_value_type / rdr_container_type (Renderer/Mesh.h:12-52)
INT8, INT16, INT32, INT64, UINT8, UINT16, UINT32, UINT64, HALF, FLOAT, DOUBLE = range(11)
SCALAR, VEC2, VEC3, VEC4 = range(4)
ATTR_POSITION, ATTR_NORMAL, ATTR_TANGENT, ATTR_TEXCOORD_0, ATTR_BITANGENT, ATTR_JOINTS_0, ATTR_WEIGHTS_0 = range(7)
ALPHA_OPAQUE, ALPHA_MASK, ALPHA_BLEND = 0, 1, 2
CAMERA_PERSPECTIVE, CAMERA_ORTHOGRAPHIC = 0, 1
POINT_LIGHT_CAST_SHADOWS, POINT_LIGHT_HAS_SHADOW_MAP, POINT_LIGHT_STATIC_SHADOW_MAP_RENDERED = 1, 2, 4
FILTER_NEAREST, FILTER_LINEAR = 0, 1
WRAP_REPEAT, WRAP_MIRROR, WRAP_CLAMP = 0, 1, 2


class Settings(C.Structure):
    _fields_ = [("hdr_output_enabled", C.c_uint32), ("dynamic_shadows_enabled", C.c_uint32),
                ("point_shadow_map_resolution", C.c_uint32), ("point_shadow_map_count", C.c_uint32),
                ("directional_shadow_map_resolution", C.c_uint32), ("directional_shadow_cascade_levels", C.c_uint32),
                ("vsync", C.c_uint32), ("exposure_bias", C.c_float), ("gamma", C.c_float), ("msaa", C.c_uint32)]


class Sampler(C.Structure):
    _fields_ = [("mag_filter", C.c_uint32), ("min_filter", C.c_uint32), ("mipmap_filter", C.c_uint32),
                ("wrap_u", C.c_uint32), ("wrap_v", C.c_uint32)]


class VertexAttribute(C.Structure):
    _fields_ = [("name", C.c_uint32), ("buffer", C.c_uint32), ("offset", C.c_uint64), ("stride", C.c_int32),
                ("container", C.c_uint32), ("count", C.c_uint64), ("value_type", C.c_uint32), ("_pad", C.c_uint32)]


class IndexBuffer(C.Structure):
    _fields_ = [("buffer", C.c_uint32), ("index_type", C.c_uint32), ("offset", C.c_uint64), ("count", C.c_uint64)]


class Material(C.Structure):
    _fields_ = [("base_color_factor", C.c_float * 4), ("emissive_factor", C.c_float * 4), ("normal_scale", C.c_float),
                ("metallic_factor", C.c_float), ("roughness_factor", C.c_float), ("ao_factor", C.c_float),
                ("alpha_mode", C.c_int32), ("alpha_cutoff", C.c_float), ("albedo_texture", C.c_uint32),
                ("normal_texture", C.c_uint32), ("metal_roughness_texture", C.c_uint32), ("ao_texture", C.c_uint32),
                ("emissive_texture", C.c_uint32), ("_pad", C.c_uint32)]


class PrimitiveDesc(C.Structure):
    _fields_ = [("attributes", C.POINTER(VertexAttribute)), ("attribute_count", C.c_uint32), ("mode", C.c_uint32),
                ("indices", IndexBuffer), ("bb_min", C.c_float * 3), ("bb_max", C.c_float * 3), ("material", Material)]


class PointLight(C.Structure):
    _fields_ = [("point", C.c_float * 3), ("intensity", C.c_float * 3), ("flags", C.c_uint32)]


class DirectionalLight(C.Structure):
    _fields_ = [("position", C.c_float * 3), ("intensity", C.c_float * 3), ("orientation_wxyz", C.c_float * 4)]


class Camera(C.Structure):
    _fields_ = [("type", C.c_int32), ("position", C.c_float * 3), ("euler", C.c_float * 3), ("near_plane", C.c_float),
                ("far_plane", C.c_float), ("fov_or_height", C.c_float), ("aspect_ratio", C.c_float)]


class FrameParams(C.Structure):
    _fields_ = [("exposure_mode", C.c_int32), ("delta_time", C.c_float), ("temporal_luminance", C.c_float),
                ("exposure", C.c_float)]


class FrameStats(C.Structure):
    _fields_ = [("triangle_instances", C.c_uint64), ("triangles_binned", C.c_uint64), ("tile_pairs", C.c_uint64),
                ("views", C.c_uint32), ("draws_main", C.c_uint32), ("ms_geometry", C.c_float), ("ms_shade", C.c_float),
                ("ms_bloom", C.c_float), ("ms_tonemap", C.c_float), ("ms_total", C.c_float), ("exposure", C.c_float),
                ("avg_log_luminance", C.c_float), ("ms_shade_kernel", C.c_float), ("ms_raster_kernel", C.c_float),
                ("chunks_culled", C.c_uint64), ("frame_index", C.c_uint64), ("blend_groups", C.c_uint32), ("host_ms", C.c_float),
                ("views_reused", C.c_uint32), ("draw_lists_on_device", C.c_uint32)]


_VP = C.c_void_p
_U32P = C.POINTER(C.c_uint32)
_FP = C.POINTER(C.c_float)
_U8P = C.POINTER(C.c_uint8)

# name -> (restype, argtypes); exactly the declarations of include/rdr.h
SIGNATURES = {
    "settings_default": (C.c_int, [C.POINTER(Settings)]),
    "create": (C.c_int, [C.POINTER(Settings), C.c_uint32, C.c_uint32, C.c_int, C.POINTER(_VP)]),
    "destroy": (None, [_VP]),
    "last_error": (C.c_char_p, [_VP]),
    "backend_name": (C.c_char_p, []),
    "load_buffer": (C.c_int, [_VP, _VP, C.c_size_t, _U32P]),
    "load_image": (C.c_int, [_VP, _VP, C.c_size_t, C.c_uint32, C.c_uint32, C.c_int, C.c_uint32, _U32P]),
    "make_texture": (C.c_int, [_VP, C.c_uint32, C.POINTER(Sampler), _U32P]),
    "add_mesh": (C.c_int, [_VP, C.POINTER(PrimitiveDesc), C.c_uint32, _FP, C.c_int, _U32P]),
    "set_mesh_transform": (C.c_int, [_VP, C.c_uint32, _FP]),
    "set_mesh_joints": (C.c_int, [_VP, C.c_uint32, _FP, C.c_uint32]),
    "set_lights": (C.c_int, [_VP, C.POINTER(PointLight), C.c_uint32, C.POINTER(DirectionalLight)]),
    "set_environment_map": (C.c_int, [_VP, C.POINTER(_FP), C.c_uint32, C.c_uint32]),
    "set_brdf_lut": (C.c_int, [_VP, _U8P, C.c_uint32, C.c_uint32, C.c_uint32]),
    "set_environment_equirect": (C.c_int, [_VP, _FP, C.c_uint32, C.c_uint32, C.c_uint32]),
    "generate_brdf_lut": (C.c_int, [_VP, C.c_uint32, C.c_uint32]),
    "read_primitive_positions": (C.c_int, [_VP, C.c_uint32, C.c_uint32, _FP, _U32P]),
    "read_brdf_lut": (C.c_int, [_VP, _U8P, _U32P, _U32P]),
    "set_camera": (C.c_int, [_VP, C.POINTER(Camera)]),
    "set_exposure_bias": (C.c_int, [_VP, C.c_float, C.c_float]),
    "render_frame": (C.c_int, [_VP, C.POINTER(FrameParams), _VP, C.c_int, _FP]),
    "synchronize": (C.c_int, [_VP]),
    "wait_frame": (C.c_int, [_VP, C.c_uint64]),
    "get_frame_stats": (C.c_int, [_VP, C.POINTER(FrameStats)]),
    "read_depth": (C.c_int, [_VP, _FP]),
    "read_visibility": (C.c_int, [_VP, _U32P]),
    "read_directional_shadow": (C.c_int, [_VP, C.c_uint32, _FP]),
    "read_point_shadow": (C.c_int, [_VP, C.c_uint32, C.c_uint32, _FP]),
    "read_hdr": (C.c_int, [_VP, _FP]),
    "keep_intermediates": (C.c_int, [_VP, C.c_uint32]),
    "read_bloom": (C.c_int, [_VP, C.c_uint32, _FP]),
    "read_log_luminance": (C.c_int, [_VP, _FP]),
    "read_env_cube": (C.c_int, [_VP, C.c_int, C.c_uint32, _FP]),
    "read_cascades": (C.c_int, [_VP, _FP, _FP]),
    "read_image_level": (C.c_int, [_VP, C.c_uint32, C.c_uint32, _U8P, _U32P, _U32P]),
    "post_process": (C.c_int, [_VP, _FP, C.c_float, _VP, _FP]),
}

# entry points only the HIP library has (the CPU checker allocates nothing on a device to begin with)
HIP_ONLY_SIGNATURES = {
    "post_process_noalloc": (C.c_int, [_VP, _VP, C.c_uint32, C.c_uint32, C.c_float, _VP, _VP, _VP]),
    "get_precompute_ms": (C.c_int, [_VP, _FP]),
}
NOALLOC_MAX_PIXELS = 1920 * 1088


class CApi:
    """All entry points of include/rdr.h, bound from `lib` under `prefix`."""

    def __init__(self, lib_path, prefix):
        self.path = str(lib_path)
        self.prefix = prefix
        self.lib = C.CDLL(self.path, mode=C.RTLD_GLOBAL)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(self.lib, prefix + name)  # AttributeError = missing export
            fn.restype = res
            fn.argtypes = args
            setattr(self, name, fn)
        if prefix == "rdr_":
            for name, (res, args) in HIP_ONLY_SIGNATURES.items():
                fn = getattr(self.lib, prefix + name)
                fn.restype = res
                fn.argtypes = args
                setattr(self, name, fn)


def fptr(a):
    return a.ctypes.data_as(_FP)


def u8ptr(a):
    return a.ctypes.data_as(_U8P)


def u32ptr(a):
    return a.ctypes.data_as(_U32P)


def as_f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)
