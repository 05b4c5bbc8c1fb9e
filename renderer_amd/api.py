"""Python mirror of the reference's public renderer surface
(Renderer/VulkanRenderer.h:94-110) on top of the C ABI of include/rdr.h.

`Renderer(capi, width, height, settings)` is backend-agnostic: `capi` is a
`_binding.CApi`. `hip_renderer()` builds one on the HIP library and raises if
that library is not built — there is no CPU fallback in this package.
"""
import ctypes as C
import os

import numpy as np

from . import _binding as B

_HERE = os.path.dirname(os.path.abspath(__file__))
# RDR_HIP_LIB selects another build of the same library (A/B experiments); there is still no non-HIP path
HIP_LIB_PATH = os.environ.get("RDR_HIP_LIB") or os.path.join(_HERE, "lib", "librdr_hip.so")


class RendererError(RuntimeError):
    pass


def default_settings(**overrides):
    """RendererSettings{} (Renderer/VulkanRenderer.h:44-59)."""
    s = B.Settings(0, 1, 512, 8, 2048, 4, 0, 0.0, 2.2, 4)
    for k, v in overrides.items():
        if not hasattr(s, k):
            raise AttributeError(k)
        setattr(s, k, v)
    return s


def load_hip_capi():
    if not os.path.exists(HIP_LIB_PATH):
        raise RendererError(
            f"{HIP_LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950). This package has no CPU path.")
    return B.CApi(HIP_LIB_PATH, "rdr_")


def quat_look_at(direction, up):
    """glm::quatLookAt (right-handed), returns (w, x, y, z) float32."""
    d = np.asarray(direction, np.float32)
    u = np.asarray(up, np.float32)
    c2 = -d
    right = np.cross(u, c2).astype(np.float32)
    c0 = (right * np.float32(1.0 / np.sqrt(max(1e-5, float(np.dot(right, right)))))).astype(np.float32)
    c1 = np.cross(c2, c0).astype(np.float32)
    m = np.stack([c0, c1, c2])  # m[col][row]
    fx = m[0][0] - m[1][1] - m[2][2]
    fy = m[1][1] - m[0][0] - m[2][2]
    fz = m[2][2] - m[0][0] - m[1][1]
    fw = m[0][0] + m[1][1] + m[2][2]
    big, fb = 0, fw
    for i, f in ((1, fx), (2, fy), (3, fz)):
        if f > fb:
            big, fb = i, f
    bv = np.float32(np.sqrt(fb + 1.0) * 0.5)
    mult = np.float32(0.25) / bv
    if big == 0:
        q = (bv, (m[1][2] - m[2][1]) * mult, (m[2][0] - m[0][2]) * mult, (m[0][1] - m[1][0]) * mult)
    elif big == 1:
        q = ((m[1][2] - m[2][1]) * mult, bv, (m[0][1] + m[1][0]) * mult, (m[2][0] + m[0][2]) * mult)
    elif big == 2:
        q = ((m[2][0] - m[0][2]) * mult, (m[0][1] + m[1][0]) * mult, bv, (m[1][2] + m[2][1]) * mult)
    else:
        q = ((m[0][1] - m[1][0]) * mult, (m[2][0] + m[0][2]) * mult, (m[1][2] + m[2][1]) * mult, bv)
    return np.array(q, np.float32)


class Renderer:
    """One renderer context (one GPU). Method names follow VulkanRenderer."""

    def __init__(self, capi, width, height, settings=None, device=0, keep_intermediates=False):
        """keep_intermediates: also store the two images nothing later in the frame reads (accumulated bloom mip 0, the
        whole log-luminance image), which `read_bloom(0)` and `read_log_luminance` need."""
        self.api = capi
        self.width, self.height = int(width), int(height)
        self.settings = settings if settings is not None else default_settings()
        self._ctx = B._VP()
        rc = capi.create(C.byref(self.settings), self.width, self.height, int(device), C.byref(self._ctx))
        if rc != B.RDR_OK:
            self._ctx = B._VP()
            raise RendererError(f"{capi.prefix}create failed with status {rc}")
        self._keep = []
        if keep_intermediates:
            self.keep_intermediates(True)

    # -- plumbing ---------------------------------------------------------
    def close(self):
        if self._ctx:
            self.api.destroy(self._ctx)
            self._ctx = B._VP()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc, what):
        if rc != B.RDR_OK:
            msg = self.api.last_error(self._ctx)
            raise RendererError(f"{what}: status {rc}: {msg.decode() if msg else ''}")

    # -- resources (VulkanRenderer.h:100-106) -------------------------------
    def load_buffer(self, data):
        a = np.ascontiguousarray(data).view(np.uint8).reshape(-1)
        out = C.c_uint32()
        self._check(self.api.load_buffer(self._ctx, a.ctypes.data_as(B._VP), a.size, C.byref(out)), "load_buffer")
        return out.value

    def load_image(self, pixels, fmt, max_mip_levels=0xFFFFFFFF):
        """pixels [h, w, channels]: uint8 for the 8-bit formats, float16 / float32 for the SFLOAT ones (Image.h:5-14)."""
        dtype = {B.FORMAT_R16G16B16_SFLOAT: np.float16, B.FORMAT_R16G16B16A16_SFLOAT: np.float16, B.FORMAT_R32G32B32_SFLOAT: np.float32,
                 B.FORMAT_R32G32B32A32_SFLOAT: np.float32}.get(int(fmt), np.uint8)
        a = np.ascontiguousarray(pixels, dtype=dtype)
        h, w = a.shape[0], a.shape[1]
        self._image_is_float = getattr(self, "_image_is_float", {})
        out = C.c_uint32()
        self._check(self.api.load_image(self._ctx, a.ctypes.data_as(B._VP), a.nbytes, w, h, int(fmt), int(max_mip_levels),
                                        C.byref(out)), "load_image")
        self._image_is_float[out.value] = dtype != np.uint8
        return out.value

    def make_texture(self, image, wrap_u=B.WRAP_REPEAT, wrap_v=B.WRAP_REPEAT, mag_filter=B.FILTER_LINEAR, min_filter=B.FILTER_LINEAR,
                     mipmap_filter=B.FILTER_LINEAR):
        s = B.Sampler(mag_filter, min_filter, mipmap_filter, wrap_u, wrap_v)
        out = C.c_uint32()
        self._check(self.api.make_texture(self._ctx, image, C.byref(s), C.byref(out)), "make_texture")
        return out.value

    def add_mesh(self, primitives, model=None, is_static=True):
        """primitives: list of dicts with keys attributes (list of VertexAttribute), indices (IndexBuffer or
        None), bb_min, bb_max, material (Material)."""
        n = len(primitives)
        descs = (B.PrimitiveDesc * n)()
        keep = []
        for i, p in enumerate(primitives):
            attrs = (B.VertexAttribute * len(p["attributes"]))(*p["attributes"])
            keep.append(attrs)
            descs[i].attributes = attrs
            descs[i].attribute_count = len(p["attributes"])
            descs[i].mode = 4
            descs[i].indices = p["indices"] if p.get("indices") is not None else B.IndexBuffer(B.INVALID_ID, B.UINT32, 0, 0)
            descs[i].bb_min = (C.c_float * 3)(*[float(x) for x in p["bb_min"]])
            descs[i].bb_max = (C.c_float * 3)(*[float(x) for x in p["bb_max"]])
            descs[i].material = p["material"]
        m = B.as_f32(np.eye(4) if model is None else model).reshape(16)
        out = C.c_uint32()
        self._check(self.api.add_mesh(self._ctx, descs, n, B.fptr(m), 1 if is_static else 0, C.byref(out)), "add_mesh")
        return out.value

    def set_mesh_transform(self, mesh_id, model):
        m = B.as_f32(model).reshape(16)
        self._check(self.api.set_mesh_transform(self._ctx, mesh_id, B.fptr(m)), "set_mesh_transform")

    def set_mesh_joints(self, mesh_id, joint_matrices):
        """joint_matrices: [n][4][4], each laid out like `model` in add_mesh (flattening gives column-major, i.e.
        the transpose of the numpy math-convention matrix)."""
        j = B.as_f32(joint_matrices).reshape(-1)
        self._check(self.api.set_mesh_joints(self._ctx, mesh_id, B.fptr(j), j.size // 16), "set_mesh_joints")

    def read_primitive_positions(self, mesh_id, primitive):
        n = C.c_uint32()
        self._check(self.api.read_primitive_positions(self._ctx, mesh_id, primitive, None, C.byref(n)), "read_primitive_positions")
        a = np.empty((n.value, 3), np.float32)
        self._check(self.api.read_primitive_positions(self._ctx, mesh_id, primitive, B.fptr(a), C.byref(n)), "read_primitive_positions")
        return a

    def set_lights(self, point_lights, directional):
        """point_lights: list of (point3, intensity3, cast_shadows); directional: (position3, intensity3, quat_wxyz)."""
        n = len(point_lights)
        arr = (B.PointLight * max(n, 1))()
        for i, (p, inten, cast) in enumerate(point_lights):
            arr[i].point = (C.c_float * 3)(*[float(x) for x in p])
            arr[i].intensity = (C.c_float * 3)(*[float(x) for x in inten])
            arr[i].flags = B.POINT_LIGHT_CAST_SHADOWS if cast else 0
        d = B.DirectionalLight()
        d.position = (C.c_float * 3)(*[float(x) for x in directional[0]])
        d.intensity = (C.c_float * 3)(*[float(x) for x in directional[1]])
        d.orientation_wxyz = (C.c_float * 4)(*[float(x) for x in directional[2]])
        self._check(self.api.set_lights(self._ctx, arr, n, C.byref(d)), "set_lights")

    def set_environment_map(self, faces):
        """faces: array [6][res][res][4] float32, order +X,-X,+Y,-Y,+Z,-Z (main.cpp:434-441)."""
        f = B.as_f32(faces)
        assert f.ndim == 4 and f.shape[0] == 6 and f.shape[3] == 4
        self._env_res = int(f.shape[1])
        ptrs = (B._FP * 6)(*[B.fptr(f[i]) for i in range(6)])
        self._check(self.api.set_environment_map(self._ctx, ptrs, f.shape[2], f.shape[1]), "set_environment_map")

    def set_environment_equirect(self, panorama, face_res):
        """panorama: [h][w][4] float32 equirectangular image; builds face_res^2 cube faces and runs the bakes."""
        p = B.as_f32(panorama)
        assert p.ndim == 3 and p.shape[2] == 4
        self._env_res = int(face_res)
        self._check(self.api.set_environment_equirect(self._ctx, B.fptr(p), p.shape[1], p.shape[0], int(face_res)), "set_environment_equirect")

    def generate_brdf_lut(self, size=400, samples=1024):
        self._check(self.api.generate_brdf_lut(self._ctx, int(size), int(samples)), "generate_brdf_lut")

    def precompute_ms(self):
        """Device time of the kernels of the last environment precompute (HIP library only): dict of ms."""
        a = np.zeros(4, np.float32)
        self._check(self.api.get_precompute_ms(self._ctx, B.fptr(a)), "get_precompute_ms")
        return dict(zip(("to_cube", "irradiance", "prefilter", "brdf_lut"), (float(v) for v in a)))

    def read_brdf_lut(self):
        w, h = C.c_uint32(), C.c_uint32()
        self._check(self.api.read_brdf_lut(self._ctx, None, C.byref(w), C.byref(h)), "read_brdf_lut")
        a = np.empty((h.value, w.value, 4), np.uint8)
        self._check(self.api.read_brdf_lut(self._ctx, B.u8ptr(a), C.byref(w), C.byref(h)), "read_brdf_lut")
        return a

    def set_brdf_lut(self, pixels):
        a = np.ascontiguousarray(pixels, dtype=np.uint8)
        self._check(self.api.set_brdf_lut(self._ctx, B.u8ptr(a), a.shape[1], a.shape[0], a.shape[2]), "set_brdf_lut")

    def set_camera(self, position, euler, near=0.1, far=60.0, fov_or_height=np.radians(40.0), aspect=None,
                   cam_type=B.CAMERA_PERSPECTIVE):
        c = B.Camera()
        c.type = cam_type
        c.position = (C.c_float * 3)(*[float(x) for x in position])
        c.euler = (C.c_float * 3)(*[float(x) for x in euler])
        c.near_plane, c.far_plane = float(near), float(far)
        c.fov_or_height = float(np.float32(fov_or_height))
        c.aspect_ratio = float(np.float32(self.width) / np.float32(self.height)) if aspect is None else float(aspect)
        self._check(self.api.set_camera(self._ctx, C.byref(c)), "set_camera")

    def set_exposure_bias(self, bias, gamma=2.2):
        self._check(self.api.set_exposure_bias(self._ctx, float(bias), float(gamma)), "set_exposure_bias")

    # -- the hot path -----------------------------------------------------
    def render_frame(self, exposure_mode=1, delta_time=1.0 / 60.0, temporal_luminance=0.25, exposure=0.0,
                     out=None, device_ptr=None):
        """Renders one frame. Returns (the output image HxWx4, uint8 or with settings.hdr_output_enabled float16, or None
        for a device destination; avg_log_luminance)."""
        fp = B.FrameParams(int(exposure_mode), float(delta_time), float(temporal_luminance), float(exposure))
        avg = C.c_float()
        if device_ptr is not None:
            self._check(self.api.render_frame(self._ctx, C.byref(fp), B._VP(int(device_ptr)), 1, None), "render_frame")
            return None, None
        if out is None:
            out = np.empty((self.height, self.width, 4), self.output_dtype)
        self._check(self.api.render_frame(self._ctx, C.byref(fp), out.ctypes.data_as(B._VP), 0, C.byref(avg)), "render_frame")
        return out, avg.value

    @property
    def output_dtype(self):
        """R8G8B8A8_UNORM, or R16G16B16A16_SFLOAT for a context created with hdr_output_enabled."""
        return np.float16 if self.settings.hdr_output_enabled else np.uint8

    def synchronize(self):
        self._check(self.api.synchronize(self._ctx), "synchronize")

    def wait_frame(self, frame_index):
        """Waits for the frame with that index (frame_stats().frame_index counts them from 1); frame_stats() then describes it."""
        self._check(self.api.wait_frame(self._ctx, int(frame_index)), "wait_frame")

    def frame_stats(self):
        s = B.FrameStats()
        self._check(self.api.get_frame_stats(self._ctx, C.byref(s)), "get_frame_stats")
        return s

    def post_process(self, hdr, exposure):
        h = B.as_f32(hdr)
        assert h.shape == (self.height, self.width, 4)
        out = np.empty((self.height, self.width, 4), self.output_dtype)
        avg = C.c_float()
        self._check(self.api.post_process(self._ctx, B.fptr(h), float(exposure), out.ctypes.data_as(B._VP), C.byref(avg)), "post_process")
        return out, avg.value

    # -- debug read-backs -------------------------------------------------
    def read_depth(self):
        a = np.empty((self.settings.msaa, self.height, self.width), np.float32)
        self._check(self.api.read_depth(self._ctx, B.fptr(a)), "read_depth")
        return a

    def read_visibility(self):
        a = np.empty((self.settings.msaa, self.height, self.width), np.uint32)
        self._check(self.api.read_visibility(self._ctx, B.u32ptr(a)), "read_visibility")
        return a

    def read_directional_shadow(self, index):
        r = self.settings.directional_shadow_map_resolution
        a = np.empty((r, r), np.float32)
        self._check(self.api.read_directional_shadow(self._ctx, index, B.fptr(a)), "read_directional_shadow")
        return a

    def read_point_shadow(self, slot, face):
        r = self.settings.point_shadow_map_resolution
        a = np.empty((r, r), np.float32)
        self._check(self.api.read_point_shadow(self._ctx, slot, face, B.fptr(a)), "read_point_shadow")
        return a

    def read_hdr(self):
        a = np.empty((self.height, self.width, 4), np.float32)
        self._check(self.api.read_hdr(self._ctx, B.fptr(a)), "read_hdr")
        return a

    def keep_intermediates(self, enable=True):
        self._check(self.api.keep_intermediates(self._ctx, 1 if enable else 0), "keep_intermediates")

    def read_bloom(self, level):
        w, h = max(1, self.width >> level), max(1, self.height >> level)
        a = np.empty((h, w, 3), np.float32)
        self._check(self.api.read_bloom(self._ctx, level, B.fptr(a)), "read_bloom")
        return a

    def read_log_luminance(self):
        a = np.empty((self.height, self.width), np.float32)
        self._check(self.api.read_log_luminance(self._ctx, B.fptr(a)), "read_log_luminance")
        return a

    def read_env_cube(self, which, level=0):
        base = {1: 256, 2: 512}.get(which)
        if base is None:
            base = self._env_res
        r = max(1, base >> level)
        a = np.empty((6, r, r, 3), np.float32)
        self._check(self.api.read_env_cube(self._ctx, which, level, B.fptr(a)), "read_env_cube")
        return a

    def read_cascades(self):
        vp = np.empty((4, 4, 4), np.float32)
        d = np.empty(4, np.float32)
        self._check(self.api.read_cascades(self._ctx, B.fptr(vp), B.fptr(d)), "read_cascades")
        return vp, d

    def read_image_level(self, image, level):
        w, h = C.c_uint32(), C.c_uint32()
        self._check(self.api.read_image_level(self._ctx, image, level, None, C.byref(w), C.byref(h)), "read_image_level")
        # as stored: RGBA8, or one float32[4] per texel for an image of an SFLOAT format
        a = np.empty((h.value, w.value, 4), np.float32 if getattr(self, "_image_is_float", {}).get(image) else np.uint8)
        self._check(self.api.read_image_level(self._ctx, image, level, a.ctypes.data_as(C.POINTER(C.c_uint8)), C.byref(w), C.byref(h)), "read_image_level")
        return a


def hip_renderer(width, height, settings=None, device=0):
    return Renderer(load_hip_capi(), width, height, settings, device)
