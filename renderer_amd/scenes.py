"""Seeded synthetic scenes (no glTF/HDR assets exist offline).

Every builder takes a `Renderer` (any backend behind the C ABI) and issues the
same load/set calls a glTF loader would (Renderer/main.cpp:341-462), so two
back ends fed by the same builder hold bit-identical inputs.

  small_scene   : parity-test scene (opaque + MASK + BLEND prims, 3 point lights)
  helmet_like   : config 1 stand-in, one ~69k-triangle textured UV sphere
  sponza_like   : config 2 stand-in, colonnaded hall, ~262k triangles, 16 point lights
  raster_stress : config 5 stand-in, displaced grid with a chosen triangle count
"""
import ctypes as C

import numpy as np

from . import _binding as B
from .api import quat_look_at

# ------------------------------------------------------------------ textures


_NOISE_CACHE = {}


def value_noise(size, seed, octaves=4, base=4):
    """Tileable value noise in [0,1], float32 [size,size] (cached, treat as read-only)."""
    key = (size, seed, octaves, base)
    if key not in _NOISE_CACHE:
        _NOISE_CACHE[key] = _value_noise(size, seed, octaves, base)
    return _NOISE_CACHE[key]


def _value_noise(size, seed, octaves, base):
    rng = np.random.default_rng(seed)
    out = np.zeros((size, size), np.float32)
    amp, total = 1.0, 0.0
    ys, xs = np.meshgrid(np.arange(size, dtype=np.float32), np.arange(size, dtype=np.float32), indexing="ij")
    for o in range(octaves):
        n = base << o
        g = rng.random((n, n), dtype=np.float32)
        fx, fy = xs * (n / size), ys * (n / size)
        x0, y0 = np.floor(fx).astype(np.int64) % n, np.floor(fy).astype(np.int64) % n
        x1, y1 = (x0 + 1) % n, (y0 + 1) % n
        tx, ty = fx - np.floor(fx), fy - np.floor(fy)
        tx, ty = tx * tx * (3 - 2 * tx), ty * ty * (3 - 2 * ty)
        v = (g[y0, x0] * (1 - tx) + g[y0, x1] * tx) * (1 - ty) + (g[y1, x0] * (1 - tx) + g[y1, x1] * tx) * ty
        out += amp * v
        total += amp
        amp *= 0.5
    return (out / total).astype(np.float32)


def to_u8(a):
    return np.clip(np.rint(a * 255.0), 0, 255).astype(np.uint8)


_TEX_CACHE = {}


def material_images(size, seed, tint=(1.0, 1.0, 1.0), metal_fraction=0.3, alpha=None):
    """The five RGBA8 images of one procedural material (albedo sRGB, normal, metalRough, ao, emissive).
    Cached per argument set so that several back ends fed the same scene share the arrays."""
    key = (size, seed, tuple(float(t) for t in tint), float(metal_fraction), None if alpha is None else (alpha.shape, float(np.sum(alpha))))
    if key not in _TEX_CACHE:
        _TEX_CACHE[key] = _material_images(size, seed, tint, metal_fraction, alpha)
    return _TEX_CACHE[key]


def material_textures(r, size, seed, tint=(1.0, 1.0, 1.0), metal_fraction=0.3, alpha=None):
    """Loads the five material textures of one procedural material; returns their texture ids."""
    imgs = material_images(size, seed, tint, metal_fraction, alpha)
    fmts = (B.FORMAT_R8G8B8A8_SRGB, B.FORMAT_R8G8B8A8_UNORM, B.FORMAT_R8G8B8A8_UNORM, B.FORMAT_R8G8B8A8_UNORM, B.FORMAT_R8G8B8A8_UNORM)
    return [r.make_texture(r.load_image(img, fmt)) for img, fmt in zip(imgs, fmts)]


def _material_images(size, seed, tint, metal_fraction, alpha):
    n0, n1, n2 = value_noise(size, seed), value_noise(size, seed + 1, base=8), value_noise(size, seed + 2, base=2)
    albedo = np.empty((size, size, 4), np.uint8)
    for k in range(3):
        albedo[..., k] = to_u8(np.clip((0.25 + 0.75 * n0) * tint[k] * (0.6 + 0.4 * n2), 0, 1))
    albedo[..., 3] = 255 if alpha is None else to_u8(alpha)
    # normal map from the height field n1
    h = n1
    dx = (np.roll(h, -1, 1) - np.roll(h, 1, 1)) * 2.0
    dy = (np.roll(h, -1, 0) - np.roll(h, 1, 0)) * 2.0
    nz = np.ones_like(h)
    ln = np.sqrt(dx * dx + dy * dy + nz * nz)
    normal = np.empty((size, size, 4), np.uint8)
    normal[..., 0] = to_u8(-dx / ln * 0.5 + 0.5)
    normal[..., 1] = to_u8(-dy / ln * 0.5 + 0.5)
    normal[..., 2] = to_u8(nz / ln * 0.5 + 0.5)
    normal[..., 3] = 255
    mr = np.empty((size, size, 4), np.uint8)
    mr[..., 0] = 0
    mr[..., 1] = to_u8(0.05 + 0.95 * n1)                        # roughness in G
    mr[..., 2] = np.where(n2 > 1.0 - metal_fraction, 255, 0)    # metallic in B
    mr[..., 3] = 255
    ao = np.empty((size, size, 4), np.uint8)
    ao[..., 0] = ao[..., 1] = ao[..., 2] = to_u8(0.6 + 0.4 * n0)
    ao[..., 3] = 255
    em = np.zeros((size, size, 4), np.uint8)
    spots = (n1 > 0.8) & (n2 > 0.6)
    em[..., 0] = np.where(spots, 255, 0)
    em[..., 1] = np.where(spots, 120, 0)
    em[..., 3] = 255
    return albedo, normal, mr, ao, em


def make_material(tex=None, base_color=(1, 1, 1, 1), emissive=(0, 0, 0), metallic=1.0, roughness=1.0, ao=1.0,
                  alpha_mode=B.ALPHA_OPAQUE, alpha_cutoff=0.5):
    m = B.Material()
    m.base_color_factor = (C.c_float * 4)(*base_color)
    m.emissive_factor = (C.c_float * 4)(emissive[0], emissive[1], emissive[2], 0.0)
    m.normal_scale, m.metallic_factor, m.roughness_factor, m.ao_factor = 1.0, metallic, roughness, ao
    m.alpha_mode, m.alpha_cutoff = alpha_mode, alpha_cutoff
    ids = tex if tex is not None else [B.INVALID_ID] * 5
    m.albedo_texture, m.normal_texture, m.metal_roughness_texture, m.ao_texture, m.emissive_texture = ids
    return m


# ------------------------------------------------------------------ geometry


class Geo:
    """Vertex arrays of one primitive: pos[n,3], nrm[n,3], tan[n,4], uv[n,2], idx[m] (uint32)."""

    def __init__(self, pos, nrm, tan, uv, idx):
        self.pos, self.nrm, self.tan, self.uv = (np.ascontiguousarray(a, np.float32) for a in (pos, nrm, tan, uv))
        self.idx = np.ascontiguousarray(idx, np.uint32).reshape(-1)

    @property
    def triangles(self):
        return self.idx.size // 3

    def transformed(self, m):
        """Bakes an affine transform into the vertices (rotation/uniform scale only for normals)."""
        m = np.asarray(m, np.float64)
        p = (self.pos.astype(np.float64) @ m[:3, :3].T + m[:3, 3]).astype(np.float32)
        rot = m[:3, :3] / np.cbrt(abs(np.linalg.det(m[:3, :3])))
        n = (self.nrm.astype(np.float64) @ rot.T).astype(np.float32)
        t = self.tan.copy()
        t[:, :3] = (self.tan[:, :3].astype(np.float64) @ rot.T).astype(np.float32)
        return Geo(p, n, t, self.uv, self.idx)


def merge(geos):
    off, idx = 0, []
    for g in geos:
        idx.append(g.idx + off)
        off += g.pos.shape[0]
    return Geo(np.concatenate([g.pos for g in geos]), np.concatenate([g.nrm for g in geos]),
               np.concatenate([g.tan for g in geos]), np.concatenate([g.uv for g in geos]), np.concatenate(idx))


def grid_indices(nu, nv):
    """Two CCW (seen from +normal when u x v = normal) triangles per quad of an (nv+1) x (nu+1) vertex grid."""
    j, i = np.meshgrid(np.arange(nv), np.arange(nu), indexing="ij")
    a = (j * (nu + 1) + i).reshape(-1)
    b, c, d = a + 1, a + nu + 1, a + nu + 2
    return np.stack([a, b, d, a, d, c], 1).reshape(-1).astype(np.uint32)


def plane(size_u, size_v, nu, nv, uv_scale=1.0):
    """Plane in XZ facing +Y, centred at the origin."""
    u, v = np.meshgrid(np.linspace(0, 1, nu + 1), np.linspace(0, 1, nv + 1), indexing="xy")
    pos = np.stack([(u - 0.5) * size_u, np.zeros_like(u), (0.5 - v) * size_v], -1).reshape(-1, 3)
    nrm = np.tile([0, 1, 0], (pos.shape[0], 1))
    tan = np.tile([1, 0, 0, 1], (pos.shape[0], 1))
    uv = np.stack([u * uv_scale, v * uv_scale], -1).reshape(-1, 2)
    return Geo(pos, nrm, tan, uv, grid_indices(nu, nv))


def uv_sphere(radius, nu, nv):
    u, v = np.meshgrid(np.linspace(0, 1, nu + 1), np.linspace(0, 1, nv + 1), indexing="xy")
    phi, theta = u * 2 * np.pi, v * np.pi
    n = np.stack([np.sin(theta) * np.cos(phi), np.cos(theta), -np.sin(theta) * np.sin(phi)], -1).reshape(-1, 3)
    t = np.stack([-np.sin(phi), np.zeros_like(phi), -np.cos(phi), np.ones_like(phi)], -1).reshape(-1, 4)
    uv = np.stack([u * 4, v * 2], -1).reshape(-1, 2)
    return Geo(n * radius, n, t, uv, grid_indices(nu, nv))


def box(sx, sy, sz, n=1, uv_scale=1.0):
    faces = []
    p = plane(1, 1, n, n, uv_scale)
    rots = [np.eye(3), np.diag([1, -1, -1.0]),
            np.array([[1, 0, 0], [0, 0, -1], [0, 1, 0.0]]), np.array([[1, 0, 0], [0, 0, 1], [0, -1, 0.0]]),
            np.array([[0, 1, 0], [-1, 0, 0], [0, 0, 1.0]]), np.array([[0, -1, 0], [1, 0, 0], [0, 0, 1.0]])]
    for rmat in rots:
        m = np.eye(4)
        m[:3, :3] = rmat
        m[:3, 3] = rmat @ np.array([0, 0.5, 0])
        faces.append(p.transformed(m))
    g = merge(faces)
    g.pos *= np.array([sx, sy, sz], np.float32)
    return g


def cylinder(radius, height, nu, nv):
    u, v = np.meshgrid(np.linspace(0, 1, nu + 1), np.linspace(0, 1, nv + 1), indexing="xy")
    phi = u * 2 * np.pi
    n = np.stack([np.cos(phi), np.zeros_like(phi), -np.sin(phi)], -1).reshape(-1, 3)
    pos = n * radius + np.stack([np.zeros_like(phi), (0.5 - v) * height, np.zeros_like(phi)], -1).reshape(-1, 3)
    t = np.stack([-np.sin(phi), np.zeros_like(phi), -np.cos(phi), np.ones_like(phi)], -1).reshape(-1, 4)
    uv = np.stack([u * 3, v * 3], -1).reshape(-1, 2)
    return Geo(pos, n, t, uv, grid_indices(nu, nv))


def translation(x, y, z):
    m = np.eye(4)
    m[:3, 3] = [x, y, z]
    return m


def rotation_x(a):
    m = np.eye(4)
    c, s = np.cos(a), np.sin(a)
    m[1, 1], m[1, 2], m[2, 1], m[2, 2] = c, -s, s, c
    return m


def rotation_y(a):
    m = np.eye(4)
    c, s = np.cos(a), np.sin(a)
    m[0, 0], m[0, 2], m[2, 0], m[2, 2] = c, s, -s, c
    return m


def upload_primitive(r, geo, material, index_type=B.UINT32):
    """One buffer per primitive laid out like a glTF .bin: [pos | nrm | tan | uv | idx]."""
    n = geo.pos.shape[0]
    idx = geo.idx.astype(np.uint16) if index_type == B.UINT16 else geo.idx
    assert index_type != B.UINT16 or n <= 65536
    parts = [geo.pos.tobytes(), geo.nrm.tobytes(), geo.tan.tobytes(), geo.uv.tobytes(), idx.tobytes()]
    offs = np.cumsum([0] + [len(p) for p in parts])
    buf = r.load_buffer(np.frombuffer(b"".join(parts), np.uint8))
    attrs = [B.VertexAttribute(B.ATTR_POSITION, buf, int(offs[0]), 12, B.VEC3, n, B.FLOAT, 0),
             B.VertexAttribute(B.ATTR_NORMAL, buf, int(offs[1]), 12, B.VEC3, n, B.FLOAT, 0),
             B.VertexAttribute(B.ATTR_TANGENT, buf, int(offs[2]), 16, B.VEC4, n, B.FLOAT, 0),
             B.VertexAttribute(B.ATTR_TEXCOORD_0, buf, int(offs[3]), 8, B.VEC2, n, B.FLOAT, 0)]
    return {"attributes": attrs, "indices": B.IndexBuffer(buf, index_type, int(offs[4]), geo.idx.size),
            "bb_min": geo.pos.min(0), "bb_max": geo.pos.max(0), "material": material}


def upload_skinned_primitive(r, geo, joints, weights, material, joint_type=B.UINT16, weight_type=B.FLOAT):
    """Like upload_primitive plus JOINTS_0 (uint8 / uint16 x4) and WEIGHTS_0 (float, or unorm8 / unorm16 x4)."""
    n = geo.pos.shape[0]
    j = np.ascontiguousarray(joints, np.uint8 if joint_type == B.UINT8 else np.uint16)
    if weight_type == B.FLOAT:
        w = np.ascontiguousarray(weights, np.float32)
    else:
        top = 255 if weight_type == B.UINT8 else 65535
        # (a normalised integer holds [0, 1]: a weight above 1 is stored as 1, not wrapped round to a small value)
        w = np.rint(np.clip(np.asarray(weights, np.float64), 0.0, 1.0) * top).astype(np.uint8 if weight_type == B.UINT8 else np.uint16)
    parts = [geo.pos.tobytes(), geo.nrm.tobytes(), geo.tan.tobytes(), geo.uv.tobytes(), j.tobytes(), w.tobytes(), geo.idx.tobytes()]
    parts = [p + b"\0" * (-len(p) % 4) for p in parts]
    offs = np.cumsum([0] + [len(p) for p in parts])
    buf = r.load_buffer(np.frombuffer(b"".join(parts), np.uint8))
    attrs = [B.VertexAttribute(B.ATTR_POSITION, buf, int(offs[0]), 12, B.VEC3, n, B.FLOAT, 0),
             B.VertexAttribute(B.ATTR_NORMAL, buf, int(offs[1]), 12, B.VEC3, n, B.FLOAT, 0),
             B.VertexAttribute(B.ATTR_TANGENT, buf, int(offs[2]), 16, B.VEC4, n, B.FLOAT, 0),
             B.VertexAttribute(B.ATTR_TEXCOORD_0, buf, int(offs[3]), 8, B.VEC2, n, B.FLOAT, 0),
             B.VertexAttribute(B.ATTR_JOINTS_0, buf, int(offs[4]), 0, B.VEC4, n, joint_type, 0),
             B.VertexAttribute(B.ATTR_WEIGHTS_0, buf, int(offs[5]), 0, B.VEC4, n, weight_type, 0)]
    return {"attributes": attrs, "indices": B.IndexBuffer(buf, B.UINT32, int(offs[6]), geo.idx.size),
            "bb_min": geo.pos.min(0), "bb_max": geo.pos.max(0), "material": material}


def bone_chain_weights(y, n_bones, length):
    """Two-bone smooth weights for points at height y of a limb whose bone k starts at k * length / n_bones."""
    seg = length / n_bones
    f = np.clip(y / seg - 0.5, 0.0, n_bones - 1.0)
    j0 = np.minimum(np.floor(f).astype(np.int64), n_bones - 2) if n_bones > 1 else np.zeros_like(f, np.int64)
    u = np.clip(f - j0, 0.0, 1.0) if n_bones > 1 else np.zeros_like(f)
    u = u * u * (3 - 2 * u)
    joints = np.zeros((y.size, 4), np.uint16)
    weights = np.zeros((y.size, 4), np.float32)
    joints[:, 0], joints[:, 1] = j0, np.minimum(j0 + 1, n_bones - 1)
    weights[:, 0], weights[:, 1] = 1.0 - u, u
    return joints, weights


class SkinnedFigure:
    """Stand-in for the skinned glTF samples (CesiumMan / Fox are not on disk): a textured limb of `n_bones` bones
    that sways under key-framed joint rotations (LINEAR slerp, one STEP and one CUBICSPLINE track), plus a second
    primitive with 8-bit joints / weights."""

    def __init__(self, r, n_bones=6, length=2.4, radius=0.22, nu=24, nv=48, tex_size=64, origin=(0.0, 0.0, 0.0), seed=61):
        from . import animation as A
        self.r, self.A = r, A
        seg = length / n_bones
        limb = cylinder(radius, length, nu, nv).transformed(translation(0, length / 2, 0))
        bulb = uv_sphere(radius * 1.6, 16, 8).transformed(translation(0, length, 0))
        self.material_args = dict(size=tex_size, seed=seed, tint=(1.0, 0.75, 0.55), metal_fraction=0.2)
        tex = material_textures(r, **self.material_args)
        prims, self.parts = [], []
        for geo, jt, wt in ((limb, B.UINT16, B.FLOAT), (bulb, B.UINT8, B.UINT8)):
            joints, weights = bone_chain_weights(geo.pos[:, 1].astype(np.float64), n_bones, length)
            self.parts.append((geo, joints, weights, jt, wt))
            prims.append(upload_skinned_primitive(r, geo, joints, weights, make_material(tex), jt, wt))
        self.model = translation(*origin)
        self.mesh = r.add_mesh(prims, model=self.model.T, is_static=False)
        # bone k: child of bone k-1, offset one segment up; rest pose = straight up from the mesh origin
        nodes = [A.Node(translation=(0, 0 if k == 0 else seg, 0), name=f"bone{k}") for k in range(n_bones)]
        for k in range(n_bones - 1):
            nodes[k].children.append(nodes[k + 1])
        self.nodes = nodes
        self.root = A.Node(matrix=self.model, children=[nodes[0]], name="figure")
        self.root.recalculate_model()
        # inverse bind matrices and keys are rounded to float32, the precision a glTF file stores them in
        inverse_bind = [np.linalg.inv(np.linalg.inv(self.model) @ n.model).astype(np.float32).astype(np.float64) for n in nodes]
        self.skin = A.Skin(nodes, inverse_bind)
        times = [0.0, 0.5, 1.0, 1.5, 2.0]
        for k, n in enumerate(nodes):
            axis = (0, 0, 1) if k % 2 == 0 else (1, 0, 0)
            amp = 0.35 if k else 0.15
            angles = [0.0, amp, 0.0, -amp, 0.0]
            keys = np.stack([A.quat_from_axis_angle(axis, a * (1 + 0.1 * k)) for a in angles]).astype(np.float32).astype(np.float64)
            if k == 2:
                n.rotation_animation = A.Animation(times, keys, A.STEP, A.REPEAT, rotation=True)
            elif k == 3:
                tang = np.zeros_like(keys)
                tang[1:-1] = (keys[2:] - keys[:-2]) / 1.0
                spline = np.stack([tang, keys, tang], 1).reshape(-1, 4)
                n.rotation_animation = A.Animation(times, spline, A.CUBICSPLINE, A.REPEAT, rotation=True)
            else:
                n.rotation_animation = A.Animation(times, keys, A.LINEAR, A.MIRROR if k == 4 else A.REPEAT, rotation=True)
        nodes[-1].scale_animation = A.Animation([0.0, 1.0, 2.0], np.float32([[1, 1, 1], [1.3, 0.8, 1.3], [1, 1, 1]]), A.LINEAR, A.REPEAT)

    def palette(self, t):
        self.root.set_animation_time(t)
        self.root.recalculate_model()
        return self.skin.palette_for_abi(self.model)

    def pose(self, t):
        self.r.set_mesh_joints(self.mesh, self.palette(t))


# ------------------------------------------------------------- environment


def sky_faces(res, seed=4, sun_dir=(0.3, 0.8, 0.5), sun_power=200.0):
    """Procedural HDR sky as six RGBA32F cube faces (+X,-X,+Y,-Y,+Z,-Z), Vulkan face orientation."""
    t = (np.arange(res, dtype=np.float32) + 0.5) / res * 2 - 1
    sc, tc = np.meshgrid(t, t, indexing="xy")
    one = np.ones_like(sc)
    dirs = [(one, -tc, -sc), (-one, -tc, sc), (sc, one, tc), (sc, -one, -tc), (sc, -tc, one), (-sc, -tc, -one)]
    sd = np.asarray(sun_dir, np.float32)
    sd /= np.linalg.norm(sd)
    noise = value_noise(res, seed, octaves=3, base=2)
    faces = np.zeros((6, res, res, 4), np.float32)
    for f, (x, y, z) in enumerate(dirs):
        ln = np.sqrt(x * x + y * y + z * z)
        x, y, z = x / ln, y / ln, z / ln
        up = np.clip(y, 0, 1)
        horizon = np.exp(-np.abs(y) * 4.0)
        col = np.stack([0.25 + 0.5 * horizon, 0.45 + 0.4 * horizon, 0.9 - 0.2 * horizon], -1) * (0.6 + 0.8 * up[..., None])
        ground = np.stack([0.18 * one, 0.16 * one, 0.14 * one], -1)
        col = np.where((y < 0)[..., None], ground * (0.5 + horizon[..., None]), col)
        cosang = x * sd[0] + y * sd[1] + z * sd[2]
        sun = sun_power * np.exp((cosang - 1.0) * 400.0) + 4.0 * np.exp((cosang - 1.0) * 12.0)
        col = col * (0.85 + 0.3 * noise[..., None]) + sun[..., None] * np.array([1.0, 0.9, 0.7], np.float32)
        faces[f, ..., :3] = col
        faces[f, ..., 3] = 1.0
    return faces


def brdf_lut(size=128, samples=128):
    """Split-sum BRDF LUT, RGB8. Column = NdotV, row 0 = roughness 1 (the orientation of the
    reference's textures/ibl_brdf_lut.png as stored on disk)."""
    ndv = (np.arange(size, dtype=np.float64) + 0.5) / size
    rough = 1.0 - (np.arange(size, dtype=np.float64) + 0.5) / size
    nv, ro = np.meshgrid(ndv, rough, indexing="xy")
    a = ro * ro
    i = np.arange(samples, dtype=np.uint32)
    bits = i.copy()
    bits = (bits << 16) | (bits >> 16)
    bits = ((bits & 0x55555555) << 1) | ((bits & 0xAAAAAAAA) >> 1)
    bits = ((bits & 0x33333333) << 2) | ((bits & 0xCCCCCCCC) >> 2)
    bits = ((bits & 0x0F0F0F0F) << 4) | ((bits & 0xF0F0F0F0) >> 4)
    bits = ((bits & 0x00FF00FF) << 8) | ((bits & 0xFF00FF00) >> 8)
    xi1, xi2 = i / samples, bits * 2.3283064365386963e-10
    vx, vz = np.sqrt(1 - nv * nv), nv
    A = np.zeros_like(nv)
    Bc = np.zeros_like(nv)
    k = a / 2.0
    for s in range(samples):
        phi = 2 * np.pi * xi1[s]
        ct = np.sqrt((1 - xi2[s]) / (1 + (a * a - 1) * xi2[s]))
        st = np.sqrt(1 - ct * ct)
        hx, hz = np.cos(phi) * st, ct
        vdh = vx * hx + vz * hz
        lz = 2 * vdh * hz - vz
        ndl, ndh, vdh = np.clip(lz, 0, 1), np.clip(hz, 0, 1), np.clip(vdh, 0, 1)
        g = (ndl / (ndl * (1 - k) + k)) * (nv / (nv * (1 - k) + k))
        gv = g * vdh / np.maximum(ndh * nv, 1e-8)
        fc = (1 - vdh) ** 5
        m = ndl > 0
        A += np.where(m, (1 - fc) * gv, 0)
        Bc += np.where(m, fc * gv, 0)
    out = np.zeros((size, size, 3), np.uint8)
    out[..., 0] = to_u8(A / samples)
    out[..., 1] = to_u8(Bc / samples)
    return out


def default_directional_light():
    """main.cpp:452"""
    d = np.array([0.05, -1.0, 0.05], np.float32)
    d /= np.float32(np.linalg.norm(d))
    return ((0.0, 100.0, 0.0), (100.0, 70.0, 20.0), quat_look_at(d, (1.0, 0.0, 0.0)))


# ------------------------------------------------------------------ scenes


def small_scene(r, tex_size=64, env_res=32, with_blend=True, dynamic_sphere=True):
    """Small room: floor, three boxes, a sphere (own dynamic node), a MASK quad and a BLEND quad."""
    t_floor = material_textures(r, tex_size, 11, tint=(0.9, 0.8, 0.7))
    t_box = material_textures(r, tex_size, 21, tint=(0.7, 0.8, 1.0), metal_fraction=0.5)
    t_sph = material_textures(r, tex_size, 31, tint=(1.0, 0.6, 0.5), metal_fraction=0.7)
    alpha = value_noise(tex_size, 77, octaves=2, base=4)
    t_mask = material_textures(r, tex_size, 41, tint=(0.5, 1.0, 0.5), alpha=(alpha > 0.5).astype(np.float32))
    t_blend = material_textures(r, tex_size, 51, tint=(0.6, 0.7, 1.0), alpha=0.3 + 0.4 * alpha)
    meshes = {}
    floor = plane(12, 12, 8, 8, uv_scale=4.0)
    meshes["floor"] = r.add_mesh([upload_primitive(r, floor, make_material(t_floor), B.UINT16)])
    boxes = [box(1.0, 1.5, 1.0, 2).transformed(translation(-1.8, 0.75, -0.5)),
             box(0.8, 0.8, 0.8, 2).transformed(translation(1.6, 0.4, 0.3) @ rotation_y(0.6)),
             box(0.5, 2.5, 0.5, 2).transformed(translation(0.3, 1.25, -2.2))]
    meshes["boxes"] = r.add_mesh([upload_primitive(r, g, make_material(t_box), B.UINT16) for g in boxes])
    sph = uv_sphere(0.6, 32, 16)
    meshes["sphere"] = r.add_mesh([upload_primitive(r, sph, make_material(t_sph), B.UINT16)],
                                  model=translation(0.0, 0.9, 0.4).T, is_static=not dynamic_sphere)
    if with_blend:
        quad = plane(1.6, 1.6, 1, 1).transformed(translation(-0.6, 0.9, 1.6) @ rotation_x(np.pi / 2))
        meshes["mask"] = r.add_mesh([upload_primitive(r, quad, make_material(t_mask, alpha_mode=B.ALPHA_MASK, alpha_cutoff=0.5), B.UINT16)])
        quad2 = plane(1.4, 1.4, 1, 1).transformed(translation(1.0, 0.9, 2.0) @ rotation_x(np.pi / 2))
        meshes["blend"] = r.add_mesh([upload_primitive(r, quad2, make_material(t_blend, alpha_mode=B.ALPHA_BLEND), B.UINT16)])
    r.set_brdf_lut(brdf_lut(64, 64))
    r.set_environment_map(sky_faces(env_res))
    r.set_camera(position=(0.4, 1.7, 5.2), euler=(-0.2, 0.05, 0.0))
    lights = [((1.5, 2.2, 1.5), (30.0, 24.0, 18.0), True), ((-2.0, 1.8, 2.0), (10.0, 14.0, 22.0), True),
              ((0.0, 3.0, -3.0), (12.0, 12.0, 12.0), False)]
    r.set_lights(lights, default_directional_light())
    return meshes


def helmet_like(r, tex_size=2048, env_res=256, nu=256, nv=135):
    """Config-1 stand-in: UV sphere 256x135 segments (69 120 triangles), 5 textures of tex_size^2."""
    tex = material_textures(r, tex_size, 1, tint=(0.9, 0.85, 0.8), metal_fraction=0.6)
    sph = uv_sphere(1.0, nu, nv)
    mid = r.add_mesh([upload_primitive(r, sph, make_material(tex, emissive=(1.0, 1.0, 1.0)))])
    r.set_brdf_lut(brdf_lut(128, 128))
    r.set_environment_map(sky_faces(env_res))
    r.set_camera(position=(0.0, 0.3, 3.2), euler=(-0.08, 0.0, 0.0))
    r.set_lights([], default_directional_light())
    return {"helmet": mid, "triangles": sph.triangles}


def sponza_like(r, tex_size=1024, env_res=256, detail=1.0, n_materials=25, seed=2):
    """Config-2 stand-in: a 30 x 12 x 15 m hall with two storeys of columns, arches, walls, floor, ceiling,
    drapes (BLEND) and foliage cards (MASK); 103 primitives, ~262k triangles at detail=1; 16 point lights on
    a 4x4 grid at y=1.15 (8 shadow casting) and the reference's directional light."""
    rng = np.random.default_rng(seed)
    mats = []
    for i in range(n_materials):
        tint = 0.5 + 0.5 * rng.random(3)
        mats.append(material_textures(r, tex_size, 100 + 10 * i, tint=tuple(tint), metal_fraction=float(rng.random() * 0.4)))

    def mat(i, **kw):
        return make_material(mats[i % n_materials], **kw)

    def seg(n):
        return max(1, int(round(n * detail)))

    prims = []  # (geo, material)
    L, Hh, Wd = 30.0, 12.0, 15.0
    prims.append((plane(L, Wd, seg(160), seg(80), uv_scale=10.0), mat(0)))                                             # floor
    prims.append((plane(L, Wd, seg(60), seg(30), uv_scale=6.0).transformed(translation(0, Hh, 0) @ rotation_x(np.pi)), mat(1)))  # ceiling
    wall_l = plane(L, Hh, seg(120), seg(48), uv_scale=6.0)
    prims.append((wall_l.transformed(translation(0, Hh / 2, -Wd / 2) @ rotation_x(np.pi / 2)), mat(2)))
    prims.append((wall_l.transformed(translation(0, Hh / 2, Wd / 2) @ rotation_x(-np.pi / 2)), mat(2)))
    wall_s = plane(Wd, Hh, seg(60), seg(48), uv_scale=6.0)
    prims.append((wall_s.transformed(translation(-L / 2, Hh / 2, 0) @ rotation_y(np.pi / 2) @ rotation_x(np.pi / 2)), mat(3)))
    prims.append((wall_s.transformed(translation(L / 2, Hh / 2, 0) @ rotation_y(-np.pi / 2) @ rotation_x(np.pi / 2)), mat(3)))
    # columns: 2 rows x 12 x 2 storeys = 48, each with base + capital boxes merged in
    k = 4
    for storey in range(2):
        for row in (-1, 1):
            for i in range(12):
                x = -L / 2 + 2.0 + i * (L - 4.0) / 11
                y0 = storey * 5.5
                col = merge([cylinder(0.35, 4.5, seg(40), seg(40)).transformed(translation(0, 2.25 + 0.5, 0)),
                             box(1.0, 0.5, 1.0, seg(4)).transformed(translation(0, 0.25, 0)),
                             box(1.0, 0.5, 1.0, seg(4)).transformed(translation(0, 5.25, 0))])
                prims.append((col.transformed(translation(x, y0, row * 4.0)), mat(k)))
                k += 1
    # arches / beams between the storeys: 2 rows x 11
    for row in (-1, 1):
        for i in range(11):
            x = -L / 2 + 2.0 + (i + 0.5) * (L - 4.0) / 11
            prims.append((box((L - 4.0) / 11 - 1.0, 0.6, 0.8, seg(10)).transformed(translation(x, 5.2, row * 4.0)), mat(k)))
            k += 1
    # upper gallery floors
    for row in (-1, 1):
        prims.append((box(L - 1.0, 0.3, 3.0, seg(30)).transformed(translation(0, 5.65, row * 5.75)), mat(k)))
        k += 1
    # drapes (BLEND) and foliage cards (MASK)
    alpha_n = value_noise(tex_size, 999, octaves=3, base=4)
    t_drape = material_textures(r, tex_size, 2001, tint=(0.9, 0.2, 0.2), alpha=0.45 + 0.3 * alpha_n)
    t_leaf = material_textures(r, tex_size, 2002, tint=(0.3, 0.9, 0.3), alpha=(alpha_n > 0.45).astype(np.float32))
    blend_prims = []
    for i in range(6):
        x = -10.0 + i * 4.0
        g = plane(2.5, 4.0, seg(20), seg(30)).transformed(translation(x, 8.0, 0.0) @ rotation_x(np.pi / 2))
        blend_prims.append((g, make_material(t_drape, alpha_mode=B.ALPHA_BLEND)))
    for i in range(16):
        x, z = -12.0 + (i % 8) * 3.4, (-1 if i < 8 else 1) * 2.2
        g = merge([plane(1.2, 1.6, seg(6), seg(8)).transformed(translation(x, 0.8, z) @ rotation_y(a) @ rotation_x(np.pi / 2)) for a in (0.0, np.pi / 2)])
        blend_prims.append((g, make_material(t_leaf, alpha_mode=B.ALPHA_MASK, alpha_cutoff=0.5)))
    # a few dynamic props
    dyn = [(uv_sphere(0.5, seg(64), seg(32)), mat(7), translation(-3.0, 1.2, 0.0)),
           (box(0.8, 0.8, 0.8, seg(8)), mat(9), translation(3.0, 0.9, 0.5) @ rotation_y(0.5))]
    tri = 0
    ids = {"static": [], "dynamic": []}
    for g, m in prims + blend_prims:
        ids["static"].append(r.add_mesh([upload_primitive(r, g, m)]))
        tri += g.triangles
    for g, m, model in dyn:
        ids["dynamic"].append(r.add_mesh([upload_primitive(r, g, m)], model=model.T, is_static=False))
        tri += g.triangles
    ids["triangles"] = tri
    ids["primitives"] = len(prims) + len(blend_prims) + len(dyn)
    r.set_brdf_lut(brdf_lut(128, 128))
    r.set_environment_map(sky_faces(env_res))
    r.set_camera(position=(-11.0, 2.2, 0.6), euler=(-0.05, -1.45, 0.0))
    lrng = np.random.default_rng(3)
    lights = []
    for j in range(4):
        for i in range(4):
            p = (-11.25 + i * 7.5, 1.15, -5.25 + j * 3.5)
            inten = tuple(20.0 + 60.0 * lrng.random(3))
            lights.append((p, inten, (i + j) % 2 == 0))
    r.set_lights(lights, default_directional_light())
    return ids


def many_primitives(r, side=100, tex_size=32, env_res=16, seed=9):
    """side x side small boxes on a floor, every one a mesh of its own with one primitive (12 triangles): what a scene
    exported object by object looks like to drawMeshes. Every 10th box is a dynamic node, every 7th blends, every
    11th is alpha-masked. The grid is mirror-symmetric about x = 0, so with a camera on that plane pairs of boxes
    have equal barycentre distances and the stable order of the draw lists shows. Returns the mesh ids by kind."""
    rng = np.random.default_rng(seed)
    tex = [material_textures(r, tex_size, 300 + 10 * i, tint=tuple(0.5 + 0.5 * rng.random(3)), metal_fraction=0.3) for i in range(4)]
    alpha = value_noise(tex_size, 78, octaves=2, base=4)
    t_mask = material_textures(r, tex_size, 341, tint=(0.5, 1.0, 0.5), alpha=(alpha > 0.45).astype(np.float32))
    t_blend = material_textures(r, tex_size, 351, tint=(0.6, 0.7, 1.0), alpha=0.3 + 0.4 * alpha)
    pitch = 0.5
    half = (side - 1) * pitch / 2
    ids = {"static": [], "dynamic": [], "floor": r.add_mesh([upload_primitive(r, plane(side * pitch + 4, side * pitch + 4, 4, 4, uv_scale=8.0),
                                                                              make_material(tex[0]), B.UINT16)])}
    unit = box(0.28, 0.28, 0.28, 1)
    k = 0
    for iz in range(side):
        for ix in range(side):
            h = 0.2 + 0.6 * ((ix * 7 + iz * 13) % 5) / 4.0
            geo = unit.transformed(translation(0.0, 0.14, 0.0))
            if k % 7 == 3:
                m = make_material(t_blend, alpha_mode=B.ALPHA_BLEND)
            elif k % 11 == 5:
                m = make_material(t_mask, alpha_mode=B.ALPHA_MASK, alpha_cutoff=0.5)
            else:
                m = make_material(tex[k % 4], base_color=(0.6 + 0.4 * ((k * 5) % 3) / 2.0, 1.0, 1.0, 1.0))
            dynamic = k % 10 == 4
            model = translation(ix * pitch - half, h if dynamic else 0.0, iz * pitch - half)
            mid = r.add_mesh([upload_primitive(r, geo, m, B.UINT16)], model=model.T, is_static=not dynamic)
            ids["dynamic" if dynamic else "static"].append((mid, ix, iz))
            k += 1
    r.set_brdf_lut(brdf_lut(32, 32))
    r.set_environment_map(sky_faces(env_res))
    r.set_camera(position=(0.0, 6.0, half + 6.0), euler=(-0.45, 0.0, 0.0))
    lights = [((0.0, 3.0, 0.0), (60.0, 50.0, 40.0), True), ((-half / 2, 2.5, half / 2), (30.0, 40.0, 60.0), True),
              ((half / 2, 4.0, -half / 2), (40.0, 40.0, 40.0), False)]
    r.set_lights(lights, default_directional_light())
    ids["pitch"], ids["half"] = pitch, half
    return ids


def raster_stress(r, quads_per_side=2236, seed=5):
    """Config-5 stand-in: displaced grid, 2 * quads_per_side^2 triangles, flat material, no textures."""
    n = quads_per_side
    g = plane(40.0, 40.0, n, n, uv_scale=8.0)
    rng = np.random.default_rng(seed)
    coarse = rng.random((33, 33)).astype(np.float32)
    u = np.linspace(0, 32, n + 1, dtype=np.float32)
    i0 = np.minimum(np.floor(u).astype(np.int64), 31)
    f = u - i0
    rows = coarse[i0] * (1 - f)[:, None] + coarse[i0 + 1] * f[:, None]
    h = rows[:, i0] * (1 - f)[None, :] + rows[:, i0 + 1] * f[None, :]
    g.pos[:, 1] = (h * 3.0).reshape(-1)
    mid = r.add_mesh([upload_primitive(r, g, make_material(base_color=(0.8, 0.7, 0.6, 1.0), metallic=0.0, roughness=0.6))])
    r.set_brdf_lut(brdf_lut(64, 64))
    r.set_environment_map(sky_faces(64))
    r.set_lights([], default_directional_light())
    return {"mesh": mid, "triangles": g.triangles}


def orbit_camera(r, k, n=8, radius=30.0, height=14.0):
    """k-th of n orbit views looking at the origin."""
    a = 2 * np.pi * k / n
    pos = np.array([radius * np.sin(a), height, radius * np.cos(a)])
    pitch = -np.arctan2(height, radius)
    r.set_camera(position=pos, euler=(pitch, a, 0.0), far=100.0)
