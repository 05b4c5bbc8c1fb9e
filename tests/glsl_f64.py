"""Float64 transcriptions of the reference's shading shaders, for known-answer tests.

Everything here is written from the GLSL text and the host lines cited next to each function, plus the Vulkan
1.3 specification for what the fixed-function hardware does (rasterisation, perspective-correct interpolation,
cube-map face selection, linear filtering, depth-compare sampling). Nothing here is derived from oracle/ or from
the HIP kernels: the point is a second, independent reading of the same source, in numpy float64.

Scope: fragments whose result does not hinge on an implementation choice. Triangle edges and shadow-compare ties are
masked out by the callers (`valid` masks). Mip-mapped material textures follow the Vulkan level-of-detail rules with
the one choice this build states for both of its back ends: isotropic filtering, with the derivatives of uv taken as
forward differences to the pixels at x + 1 and y + 1 on the fragment's own triangle.
"""
import numpy as np

PI = np.pi

# ------------------------------------------------------------------ matrices (math convention: M[row, col], column vectors)


def rotate(angle, axis):
    """glm::rotate(angle, axis) for a unit coordinate axis."""
    c, s = np.cos(angle), np.sin(angle)
    m = np.eye(4)
    if axis == 0:
        m[1, 1], m[1, 2], m[2, 1], m[2, 2] = c, -s, s, c
    elif axis == 1:
        m[0, 0], m[0, 2], m[2, 0], m[2, 2] = c, s, -s, c
    else:
        m[0, 0], m[0, 1], m[1, 0], m[1, 1] = c, -s, s, c
    return m


def translate(v):
    m = np.eye(4)
    m[:3, 3] = v
    return m


def perspective_rh_zo(fovy, aspect, n, f):
    """glm::perspective with GLM_FORCE_DEPTH_ZERO_TO_ONE (Renderer.vcxproj:92), right-handed."""
    t = np.tan(fovy / 2.0)
    m = np.zeros((4, 4))
    m[0, 0] = 1.0 / (aspect * t)
    m[1, 1] = 1.0 / t
    m[2, 2] = f / (n - f)
    m[3, 2] = -1.0
    m[2, 3] = -(f * n) / (f - n)
    return m


def ortho_rh_zo(l, r, b, t, n, f):
    """glm::ortho with GLM_FORCE_DEPTH_ZERO_TO_ONE, right-handed."""
    m = np.eye(4)
    m[0, 0] = 2.0 / (r - l)
    m[1, 1] = 2.0 / (t - b)
    m[2, 2] = -1.0 / (f - n)
    m[0, 3] = -(r + l) / (r - l)
    m[1, 3] = -(t + b) / (t - b)
    m[2, 3] = -n / (f - n)
    return m


def camera_view_proj(position, euler, near, far, fov_or_height, aspect, orthographic=False):
    """Camera.cpp:136-139 (view = Rx(-ex) Ry(-ey) Rz(-ez) T(-p)), :196-206 (proj = scale(1,-1,1) * glm proj),
    :208-216 (viewProj = crop * proj * view, crop = identity for the main camera)."""
    ex, ey, ez = (float(e) for e in euler)
    view = rotate(-ex, 0) @ rotate(-ey, 1) @ rotate(-ez, 2) @ translate(-np.asarray(position, np.float64))
    if orthographic:
        h, a = fov_or_height, aspect
        proj = ortho_rh_zo(-h * a / 2, h * a / 2, -h, h, near, far)
    else:
        proj = perspective_rh_zo(fov_or_height, aspect, near, far)
    flip = np.diag([1.0, -1.0, 1.0, 1.0])
    return flip @ proj @ view


def quat_rotate(q_wxyz, v):
    """glm: quat * vec3."""
    w, x, y, z = (float(c) for c in q_wxyz)
    qv = np.array([x, y, z])
    v = np.asarray(v, np.float64)
    uv = np.cross(qv, v)
    uuv = np.cross(qv, uv)
    return v + ((uv * w) + uuv) * 2.0


def normalize(v):
    return v / np.sqrt(np.sum(v * v, axis=-1, keepdims=True))


def dot(a, b):
    return np.sum(a * b, axis=-1)


# ------------------------------------------------------------------ storage formats


def srgb_to_linear(u8):
    v = np.asarray(u8, np.float64) / 255.0
    return np.where(v <= 0.04045, v / 12.92, ((v + 0.055) / 1.055) ** 2.4)


def to_half(x):
    """Round to nearest even binary16 and back (the R16G16B16A16_SFLOAT colour attachment, VulkanRenderer.cpp:556-572)."""
    with np.errstate(over="ignore"):
        return np.asarray(x, np.float64).astype(np.float16).astype(np.float64)


# ------------------------------------------------------------------ cube maps (Vulkan 1.3, "Cube Map Face Selection")


def cube_face_coords(d):
    """Direction(s) [...,3] -> (face, s, t). Faces +X,-X,+Y,-Y,+Z,-Z; z wins ties over y over x."""
    d = np.asarray(d, np.float64)
    rx, ry, rz = d[..., 0], d[..., 1], d[..., 2]
    ax, ay, az = np.abs(rx), np.abs(ry), np.abs(rz)
    is_z = (az >= ax) & (az >= ay)
    is_y = ~is_z & (ay >= ax)
    is_x = ~is_z & ~is_y
    face = np.where(is_z, np.where(rz < 0, 5, 4), np.where(is_y, np.where(ry < 0, 3, 2), np.where(rx < 0, 1, 0)))
    sc = np.select([face == 0, face == 1, face == 2, face == 3, face == 4, face == 5], [-rz, rz, rx, rx, rx, -rx])
    tc = np.select([face == 0, face == 1, face == 2, face == 3, face == 4, face == 5], [-ry, -ry, rz, -rz, -ry, -ry])
    ma = np.where(is_z, az, np.where(is_y, ay, ax))
    return face, 0.5 * (sc / ma + 1.0), 0.5 * (tc / ma + 1.0)


def cube_face_point(face, sc, tc):
    """Inverse of the table above: the point on the unit cube (major component +-1) with face coordinates (sc, tc)."""
    one = np.ones_like(sc)
    pts = [np.stack(p, -1) for p in ((one, -tc, -sc), (-one, -tc, sc), (sc, one, tc), (sc, -one, -tc), (sc, -tc, one), (-sc, -tc, -one))]
    out = np.zeros(np.shape(sc) + (3,))
    for f in range(6):
        out = np.where((face == f)[..., None], pts[f], out)
    return out


def cube_fetch(cube, face, i, j):
    """Texel (i, j) of `face`, i or j possibly one step outside the face: seamless cube maps take the texel across
    the shared edge. Done geometrically: the texel centre that sticks out by e beyond the edge is laid onto the
    neighbouring face, e inwards from the edge at the same position along it. Returns (texels, ok); ok is False at
    cube corners (both indices outside), where implementations differ."""
    r = cube.shape[1]
    out_i, out_j = (i < 0) | (i >= r), (j < 0) | (j >= r)
    sc, tc = (2.0 * i + 1.0) / r - 1.0, (2.0 * j + 1.0) / r - 1.0
    e = np.maximum(np.abs(sc) - 1.0, 0.0) + np.maximum(np.abs(tc) - 1.0, 0.0)
    normal = cube_face_point(face, np.zeros_like(sc), np.zeros_like(tc))
    p = cube_face_point(face, np.clip(sc, -1.0, 1.0), np.clip(tc, -1.0, 1.0)) - e[..., None] * normal
    f2, s2, t2 = cube_face_coords(p)
    i2 = np.clip(np.floor(s2 * r), 0, r - 1).astype(np.int64)
    j2 = np.clip(np.floor(t2 * r), 0, r - 1).astype(np.int64)
    return cube[f2, j2, i2], ~(out_i & out_j)


def cube_sample(cube, d):
    """texture(samplerCube, d) on one mip level, LINEAR filter. cube: [6, r, r, C]. Returns (value, ok)."""
    r = cube.shape[1]
    face, s, t = cube_face_coords(d)
    u, v = s * r - 0.5, t * r - 0.5
    i0, j0 = np.floor(u), np.floor(v)
    a, b = (u - i0)[..., None], (v - j0)[..., None]
    i0, j0 = i0.astype(np.int64), j0.astype(np.int64)
    c00, k00 = cube_fetch(cube, face, i0, j0)
    c10, k10 = cube_fetch(cube, face, i0 + 1, j0)
    c01, k01 = cube_fetch(cube, face, i0, j0 + 1)
    c11, k11 = cube_fetch(cube, face, i0 + 1, j0 + 1)
    val = (1 - b) * ((1 - a) * c00 + a * c10) + b * ((1 - a) * c01 + a * c11)
    return val, k00 & k10 & k01 & k11


def cube_sample_lod(levels, d, lod):
    """textureLod(samplerCube, d, lod) with a LINEAR mipmap mode; levels: list of [6, r_l, r_l, C]."""
    lod = np.clip(lod, 0.0, len(levels) - 1.0)
    l0 = np.floor(lod).astype(np.int64)
    l1 = np.minimum(l0 + 1, len(levels) - 1)
    f = (lod - l0)[..., None]
    out = np.zeros(np.shape(lod) + (levels[0].shape[-1],))
    ok = np.ones(np.shape(lod), bool)
    for lv in range(len(levels)):
        m0, m1 = l0 == lv, l1 == lv
        if not (m0.any() or m1.any()):
            continue
        val, k = cube_sample(levels[lv], d)
        out = out + np.where(m0[..., None], (1 - f) * val, 0.0) + np.where(m1[..., None], f * val, 0.0)
        ok &= k | ~(m0 | (m1 & (f[..., 0] > 0)))
    return out, ok


def texture_2d_repeat(img, u, v):
    """texture(sampler2D) on a single-level UNORM image [h, w, C] with LINEAR filter and REPEAT addressing."""
    h, w = img.shape[:2]
    x, y = u * w - 0.5, v * h - 0.5
    i0, j0 = np.floor(x), np.floor(y)
    a, b = (x - i0)[..., None], (y - j0)[..., None]
    i0, j0 = i0.astype(np.int64), j0.astype(np.int64)
    t = np.asarray(img, np.float64) / 255.0
    c00, c10 = t[j0 % h, i0 % w], t[j0 % h, (i0 + 1) % w]
    c01, c11 = t[(j0 + 1) % h, i0 % w], t[(j0 + 1) % h, (i0 + 1) % w]
    return (1 - b) * ((1 - a) * c00 + a * c10) + b * ((1 - a) * c01 + a * c11)


def shadow_compare_linear(depth_map, s, t, ref, eps=2e-6):
    """sampler2DShadow with LINEAR filter, compare LESS_OR_EQUAL, CLAMP_TO_BORDER with an opaque-white border
    (VulkanRendererShadow.cpp:73): each of the four texels is compared with `ref` and the 0/1 results are filtered.
    Returns (value, robust); robust is False where a comparison is decided by less than `eps`."""
    res = depth_map.shape[0]
    u, v = s * res - 0.5, t * res - 0.5
    i0, j0 = np.floor(u), np.floor(v)
    a, b = u - i0, v - j0
    i0, j0 = i0.astype(np.int64), j0.astype(np.int64)

    def tap(i, j):
        inside = (i >= 0) & (i < res) & (j >= 0) & (j < res)
        d = np.where(inside, depth_map[np.clip(j, 0, res - 1), np.clip(i, 0, res - 1)].astype(np.float64), 1.0)
        return (ref <= d).astype(np.float64), np.abs(ref - d) > eps

    r00, k00 = tap(i0, j0)
    r10, k10 = tap(i0 + 1, j0)
    r01, k01 = tap(i0, j0 + 1)
    r11, k11 = tap(i0 + 1, j0 + 1)
    val = (1 - b) * ((1 - a) * r00 + a * r10) + b * ((1 - a) * r01 + a * r11)
    return val, k00 & k10 & k01 & k11


def wrap_repeat(i, n):
    return np.mod(i, n)


REPEAT, MIRRORED_REPEAT, CLAMP_TO_EDGE = 0, 1, 2


def wrap_texel(i, n, mode):
    """Vulkan 1.3 "Wrapping Operation" on an integer texel coordinate: REPEAT i mod n; MIRRORED_REPEAT
    (n - 1) - mirror((i mod 2n) - n) with mirror(k) = k if k >= 0 else -(1 + k); CLAMP_TO_EDGE clamp(i, 0, n - 1)."""
    if mode == REPEAT:
        return np.mod(i, n)
    if mode == MIRRORED_REPEAT:
        k = np.mod(i, 2 * n) - n
        return (n - 1) - np.where(k >= 0, k, -(1 + k))
    return np.clip(i, 0, n - 1)


def reference_address_modes(wrap_u, wrap_v):
    """The address modes VulkanRenderer.cpp:370-371 builds from a Sampler's wrapU / wrapV: U as named; V REPEAT if wrapV
    says REPEAT, otherwise MIRRORED_REPEAT if *wrapU* is the mirror mode and CLAMP_TO_EDGE if it is not (the second
    line of the reference tests wrapU where one would expect wrapV)."""
    return wrap_u, (REPEAT if wrap_v == REPEAT else (MIRRORED_REPEAT if wrap_u == MIRRORED_REPEAT else CLAMP_TO_EDGE))


def bilinear_repeat(level, u, v, modes=(REPEAT, REPEAT)):
    """LINEAR filter on one level [h, w, C] of already decoded float texels with the address modes `modes` = (U, V)."""
    h, w = level.shape[:2]
    x, y = u * w - 0.5, v * h - 0.5
    i0, j0 = np.floor(x), np.floor(y)
    a, b = (x - i0)[..., None], (y - j0)[..., None]
    i0, j0 = i0.astype(np.int64), j0.astype(np.int64)
    iu0, iu1 = wrap_texel(i0, w, modes[0]), wrap_texel(i0 + 1, w, modes[0])
    jv0, jv1 = wrap_texel(j0, h, modes[1]), wrap_texel(j0 + 1, h, modes[1])
    c00, c10 = level[jv0, iu0], level[jv0, iu1]
    c01, c11 = level[jv1, iu0], level[jv1, iu1]
    return (1 - b) * ((1 - a) * c00 + a * c10) + b * ((1 - a) * c01 + a * c11)


def decode_texels(u8, srgb):
    """RGBA8 bytes -> float texels as the sampler sees them: sRGB images decode R, G, B before filtering (alpha is
    always linear), UNORM images are byte / 255."""
    if np.asarray(u8).dtype != np.uint8:
        return np.asarray(u8, np.float64)  # a level of an SFLOAT image: the sampler sees the stored values
    t = np.asarray(u8, np.float64) / 255.0
    if srgb:
        t = t.copy()
        t[..., :3] = srgb_to_linear(np.asarray(u8)[..., :3])
    return t


def texture_lod(levels, dudx, dvdx, dudy, dvdy):
    """Vulkan 1.3 "Level-of-Detail Operation" for an isotropic sampler without bias or clamp: rho = the longer of the
    two derivative vectors in texels of level 0, lambda = log2(rho), clamped to the chain."""
    h0, w0 = levels[0].shape[:2]
    rho = np.maximum(np.hypot(dudx * w0, dvdx * h0), np.hypot(dudy * w0, dvdy * h0))
    with np.errstate(divide="ignore"):
        lam = np.log2(rho)
    return np.clip(lam, 0.0, len(levels) - 1.0)


FILTER_NEAREST, FILTER_LINEAR = 0, 1


def nearest_texel(level, u, v, modes=(REPEAT, REPEAT), eps=0.02):
    """NEAREST filter on one level (Vulkan 1.3 "Texel Filtering": i = floor(u) on the unnormalised coordinate, then the
    wrapping operation). Returns (texels, robust): robust is False within `eps` texels of a texel boundary."""
    h, w = level.shape[:2]
    x, y = u * w, v * h
    i, j = np.floor(x), np.floor(y)
    fx, fy = x - i, y - j
    ok = (np.minimum(fx, 1 - fx) > eps) & (np.minimum(fy, 1 - fy) > eps)
    return level[wrap_texel(j.astype(np.int64), h, modes[1]), wrap_texel(i.astype(np.int64), w, modes[0])], ok


def texture_2d(levels, u, v, dudx, dvdx, dudy, dvdy, modes=(REPEAT, REPEAT), filters=(FILTER_LINEAR, FILTER_LINEAR, FILTER_LINEAR)):
    """texture(sampler2D, uv) with the address modes `modes` and the Sampler's `filters` = (magFilter, minFilter,
    mipmapFilter) as VulkanRenderer.cpp:363-385 hands them to the driver. Vulkan 1.3: lambda = log2(rho) clamped to the
    chain; the filter within a level is magFilter where lambda <= 0 and minFilter above; mipmap mode LINEAR blends the
    two levels around lambda by its fraction, NEAREST takes the level ceil(lambda + 0.5) - 1. `levels`: decoded float
    texels per mip level. Returns (value [n, C], lambda [n], robust [n]): robust is False where a discontinuous choice
    (filter switch, level rounding, nearest texel) is decided by a margin a float32 implementation may not resolve."""
    mag, mnf, mip = filters
    q = len(levels) - 1
    h0, w0 = levels[0].shape[:2]
    rho = np.maximum(np.hypot(dudx * w0, dvdx * h0), np.hypot(dudy * w0, dvdy * h0))
    with np.errstate(divide="ignore"):
        raw = np.log2(rho)
    lam = np.clip(raw, 0.0, float(q))
    robust = np.ones(np.shape(u), bool)
    minified = raw > 0.0
    if mag != mnf:
        robust &= np.abs(raw) > 0.02
    use_nearest = np.where(minified, mnf == FILTER_NEAREST, mag == FILTER_NEAREST)
    if mip == FILTER_NEAREST:
        d = np.maximum(np.ceil(lam + 0.5) - 1.0, 0.0)
        inner = (lam > 0.0) & (lam < q)
        robust &= ~inner | (np.abs(lam + 0.5 - np.round(lam + 0.5)) > 0.02)
        l0 = l1 = d.astype(np.int64)
        f = np.zeros(np.shape(u))[..., None]
    else:
        l0 = np.floor(lam).astype(np.int64)
        l1 = np.minimum(l0 + 1, q)
        f = (lam - l0)[..., None]
    out = np.zeros(np.shape(u) + (levels[0].shape[-1],))
    for lv in range(len(levels)):
        m0, m1 = l0 == lv, (l1 == lv) & (f[..., 0] > 0.0)
        if not (m0.any() or m1.any()):
            continue
        val = bilinear_repeat(levels[lv], u, v, modes)
        if use_nearest.any():
            near, ok = nearest_texel(levels[lv], u, v, modes)
            val = np.where(use_nearest[..., None], near, val)
            robust &= ~(use_nearest & (m0 | m1)) | ok
        out = out + np.where(m0[..., None], (1 - f) * val, 0.0) + np.where(m1[..., None], f * val, 0.0)
    return out, lam, robust


def blit_linear(src_u8, dw, dh, srgb):
    """vkCmdBlitImage with VK_FILTER_LINEAR from a whole level to the whole next one (VulkanRenderer.cpp:311-346):
    destination texel (i, j) samples the source at ((i + 0.5) * sw / dw, (j + 0.5) * sh / dh) in unnormalised
    coordinates with clamp-to-edge; sRGB formats filter in linear light and encode again. Returns the unrounded
    result in units of one byte step, [dh, dw, 4] float64 (what the stored byte must round from)."""
    sh, sw = src_u8.shape[:2]
    t = decode_texels(src_u8, srgb)
    i, j = np.meshgrid(np.arange(dw), np.arange(dh), indexing="xy")
    x, y = (i + 0.5) * (sw / dw) - 0.5, (j + 0.5) * (sh / dh) - 0.5
    i0, j0 = np.floor(x), np.floor(y)
    a, b = (x - i0)[..., None], (y - j0)[..., None]
    i0, j0 = i0.astype(np.int64), j0.astype(np.int64)
    ci = lambda k: np.clip(k, 0, sw - 1)  # noqa: E731
    cj = lambda k: np.clip(k, 0, sh - 1)  # noqa: E731
    val = (1 - b) * ((1 - a) * t[cj(j0), ci(i0)] + a * t[cj(j0), ci(i0 + 1)]) + b * ((1 - a) * t[cj(j0 + 1), ci(i0)] + a * t[cj(j0 + 1), ci(i0 + 1)])
    out = val * 255.0
    if srgb:
        rgb = np.clip(val[..., :3], 0.0, 1.0)
        out[..., :3] = np.where(rgb <= 0.0031308, rgb * 12.92, 1.055 * rgb ** (1 / 2.4) - 0.055) * 255.0
    return out


def blit_linear_float(src, dw, dh):
    """blit_linear for a level of float texels [sh, sw, C] (float64 in, float64 out, no encoding)."""
    sh, sw = src.shape[:2]
    i, j = np.meshgrid(np.arange(dw), np.arange(dh), indexing="xy")
    x, y = (i + 0.5) * (sw / dw) - 0.5, (j + 0.5) * (sh / dh) - 0.5
    i0, j0 = np.floor(x), np.floor(y)
    a, b = (x - i0)[..., None], (y - j0)[..., None]
    i0, j0 = i0.astype(np.int64), j0.astype(np.int64)
    ci = lambda k: np.clip(k, 0, sw - 1)  # noqa: E731
    cj = lambda k: np.clip(k, 0, sh - 1)  # noqa: E731
    return (1 - b) * ((1 - a) * src[cj(j0), ci(i0)] + a * src[cj(j0), ci(i0 + 1)]) + b * ((1 - a) * src[cj(j0 + 1), ci(i0)] + a * src[cj(j0 + 1), ci(i0 + 1)])


def cube_shadow_compare(depth_cube, d, ref, eps=2e-6):
    """texture(samplerCubeArrayShadow, vec4(d, layer), ref) for one layer: LINEAR filter, compare LESS_OR_EQUAL
    (VulkanRendererShadow.cpp:73 is the one shadow sampler, used for both arrays). Face and (s, t) by the cube rules,
    the four texels of the footprint (across face edges where it leaves the face: seamless cube maps) are each compared
    with `ref`, the 0 / 1 results are filtered. depth_cube: [6, r, r]. Returns (value, robust): robust is False where a
    comparison is decided by less than `eps` or the footprint touches a cube corner."""
    cube = np.asarray(depth_cube, np.float64)[..., None]
    r = cube.shape[1]
    face, s, t = cube_face_coords(d)
    u, v = s * r - 0.5, t * r - 0.5
    i0, j0 = np.floor(u), np.floor(v)
    a, b = u - i0, v - j0
    i0, j0 = i0.astype(np.int64), j0.astype(np.int64)
    res, ok = [], np.ones(np.shape(ref), bool)
    for di, dj in ((0, 0), (1, 0), (0, 1), (1, 1)):
        dep, k = cube_fetch(cube, face, i0 + di, j0 + dj)
        dep = dep[..., 0]
        res.append((ref <= dep).astype(np.float64))
        ok &= k & (np.abs(ref - dep) > eps)
    val = (1 - b) * ((1 - a) * res[0] + a * res[1]) + b * ((1 - a) * res[2] + a * res[3])
    return val, ok


# ------------------------------------------------------------------ pbr.frag


def distribution_ggx(N, H, roughness):  # pbr.frag:232-244
    a = roughness * roughness
    a2 = a * a
    ndh = np.clip(dot(N, H), 0.0, 1.0)
    denom = ndh * ndh * (a2 - 1.0) + 1.0
    return a2 / (PI * denom * denom)


def geometry_schlick_ggx(ndv, roughness):  # pbr.frag:246-255
    r = roughness + 1.0
    k = (r * r) / 8.0
    return ndv / (ndv * (1.0 - k) + k)


def geometry_smith(N, V, L, roughness):  # pbr.frag:256-264
    ndv = np.maximum(dot(N, V), 0.001)
    ndl = np.maximum(dot(N, L), 0.001)
    return geometry_schlick_ggx(ndl, roughness) * geometry_schlick_ggx(ndv, roughness)


def fresnel_schlick(cos_theta, F0):  # pbr.frag:266-269
    return F0 + (1.0 - F0) * (1.0 - np.clip(cos_theta, 0.0, 1.0))[..., None] ** 5.0


def brdf(radiance, L, N, V, albedo, roughness, metallic, F0):  # pbr.frag:211-230
    H = normalize(V + L)
    ndf = distribution_ggx(N, H, roughness)
    g = geometry_smith(N, V, L, roughness)
    F = fresnel_schlick(np.maximum(dot(H, V), 0.0), F0)
    kD = (1.0 - F) * np.asarray(1.0 - metallic)[..., None]
    numerator = (ndf * g)[..., None] * F
    denominator = np.maximum(4.0 * np.maximum(dot(N, V), 1e-3) * np.maximum(dot(N, L), 1e-3), 1e-5)
    specular = numerator / denominator[..., None]
    ndl = np.maximum(dot(N, L), 0.0)
    return (kD * albedo / PI + specular) * radiance * ndl[..., None]


POINT_SHADOW_NEAR, POINT_SHADOW_FAR = 0.1, 100.0  # pbr.frag:90-91


def material_samples(frag, mat):
    """The five texture(sampler, uv) values of pbr.frag:95-108 for n fragments, [n, 4] each, decoded (sRGB for the base
    colour only, main.cpp:138). A material either carries 1x1 texels (`albedo`, `normal`, ... as uint8[4]) or mip
    chains (`chains`: name -> list of uint8 levels) sampled at the fragments' uv with their derivatives."""
    n = frag["world"].shape[0]
    names = ("albedo", "normal", "metal_rough", "ao_tex", "emissive_tex")
    out = {"robust": np.ones(n, bool)}
    for name in names:
        srgb = name == "albedo"
        if mat.get("chains"):
            levels = [decode_texels(lv, srgb) for lv in mat["chains"][name]]
            out[name], _, ok = texture_2d(levels, frag["uv"][:, 0], frag["uv"][:, 1], frag["duvdx"][:, 0], frag["duvdx"][:, 1],
                                          frag["duvdy"][:, 0], frag["duvdy"][:, 1], reference_address_modes(*mat.get("wrap", (REPEAT, REPEAT))),
                                          mat.get("filters", (FILTER_LINEAR, FILTER_LINEAR, FILTER_LINEAR)))
            out["robust"] = out["robust"] & ok
        else:
            out[name] = np.broadcast_to(decode_texels(np.asarray(mat[name], np.uint8), srgb), (n, 4))
    return out


def pbr_frag(frag, mat, scene):
    """pbr.frag:93-209 for n fragments.

    frag: dict of interpolated vertex outputs, each [n, ...]: world (3), normal (3), tangent (3), bitangent (3),
          ls (n_cascades, 3) = directionalLightSpaceCoords; for mip-mapped materials also uv (2) and its forward
          differences duvdx, duvdy (2 each).
    mat:  dict: base_color (4), emissive (3), metallic, roughness, ao, alpha_mode, alpha_cutoff and its five textures
          (see material_samples); optionally wrap (wrapU, wrapV) and filters (mag, min, mipmap) of its Sampler.
    scene: dict: cam (3), point_lights [(pos, intensity, cube)] where cube is None for a light without a shadow map
          and its six depth faces [6, r, r] otherwise, dir_direction (3), dir_intensity (3),
          cascades: list of depth maps, prefiltered: list of cube levels, irradiance: cube, lut: uint8 [h, w, >=2].
    Returns (rgba [n, 4], robust [n], info) with info["point_shadow"][i] = the shadow factor of light i per fragment
    (NaN where the light was skipped before its look-up)."""
    n = frag["world"].shape[0]
    tex = material_samples(frag, mat)
    robust = tex["robust"].copy()
    albedo = tex["albedo"] * np.asarray(mat["base_color"], np.float64)                      # :95
    masked = np.zeros(n, bool)
    if mat["alpha_mode"] == 1:                                                              # :96-99
        masked = albedo[:, 3] < mat["alpha_cutoff"]
        robust &= np.abs(albedo[:, 3] - mat["alpha_cutoff"]) > 1e-4
    metallic = np.clip(tex["metal_rough"][:, 2] * mat["metallic"], 0.0, 1.0)                # :100-101
    roughness = np.clip(tex["metal_rough"][:, 1] * mat["roughness"], 0.05, 1.0)             # :102
    ao = tex["ao_tex"][:, 0] * mat["ao"]                                                    # :103
    emissive = tex["emissive_tex"][:, :3] * np.asarray(mat["emissive"], np.float64)         # :104
    T, Bt, Nn = normalize(frag["tangent"]), normalize(frag["bitangent"]), normalize(frag["normal"])  # :106
    nt = tex["normal"][:, :3] * 2.0 - 1.0                                                   # :108-109
    N = normalize(T * nt[:, 0:1] + Bt * nt[:, 1:2] + Nn * nt[:, 2:3])                       # :110
    P = frag["world"]
    V = normalize(np.asarray(scene["cam"], np.float64) - P)                                 # :112
    I = -V
    Rv = I - 2.0 * dot(N, I)[..., None] * N                                                 # :114 reflect(-V, N)
    alb = albedo[:, :3]
    F0 = 0.04 * (1.0 - metallic)[:, None] + alb * metallic[:, None]                         # :116-117 mix
    Lo = np.zeros((n, 3))
    info = {"point_shadow": []}
    for light in scene["point_lights"]:                                                     # :123-160
        lpos, lint = np.asarray(light[0], np.float64), np.asarray(light[1], np.float64)
        cube = light[2] if len(light) > 2 else None
        d = np.sqrt(np.sum((lpos - P) ** 2, -1))
        radiance = lint / (d * d)[..., None]
        peak = radiance.max(-1)
        on = peak > 1e-3                                                                    # :133
        robust &= np.abs(peak - 1e-3) > 1e-6
        L = lpos - P                                                                        # :137
        if cube is None:                                                                    # :157-158: L stays un-normalised
            Lo += np.where(on[..., None], brdf(radiance, L, N, V, alb, roughness, metallic, F0), 0.0)
            info["point_shadow"].append(np.full(n, np.nan))
            continue
        light_depth = np.abs(L).max(-1) - 0.05                                              # :139-141
        far, near = POINT_SHADOW_FAR, POINT_SHADOW_NEAR
        light_depth = (far - far * near / light_depth) / (far - near)                       # :143
        shadow, ok = cube_shadow_compare(cube, -L, light_depth)                             # :149
        lit = on & ~(shadow < 1e-3)                                                         # :151-152
        robust &= ~on | (ok & (np.abs(shadow - 1e-3) > 1e-6))
        Ln = normalize(L)                                                                   # :154
        Lo += np.where(lit[..., None], shadow[..., None] * brdf(radiance, Ln, N, V, alb, roughness, metallic, F0), 0.0)  # :155
        info["point_shadow"].append(np.where(on, shadow, np.nan))
    if scene.get("dir_direction") is not None:                                              # :163-183
        ls = frag["ls"]
        nc = ls.shape[1]
        split = np.full(n, nc)
        for k in range(nc - 1, -1, -1):  # first cascade (lowest k) that contains the point wins
            c = ls[:, k]
            lo_, hi_ = np.minimum(c[:, 0], c[:, 1]), np.maximum(c[:, 0], c[:, 1])
            inside = (lo_ > 1e-2) & (hi_ < 1.0 - 1e-2)
            robust &= (np.abs(lo_ - 1e-2) > 1e-5) & (np.abs(hi_ - (1.0 - 1e-2)) > 1e-5)
            split = np.where(inside, k, split)
        # the loop leaves lsCoords at the last cascade tried: the chosen one, or the last when none matched
        sel = np.minimum(split, nc - 1)
        coords = ls[np.arange(n), sel]
        shadow = np.zeros(n)
        for k in range(nc):
            m = sel == k  # a layer index past the end clamps to the last layer
            if not m.any():
                continue
            val, ok = shadow_compare_linear(scene["cascades"][k], coords[m, 0], coords[m, 1], coords[m, 2])
            shadow[m] = val
            robust[m] &= ok
        robust &= np.abs(shadow - 1e-3) > 1e-6
        L = np.broadcast_to(normalize(np.asarray(scene["dir_direction"], np.float64)), (n, 3))
        contrib = shadow[..., None] * brdf(np.asarray(scene["dir_intensity"], np.float64), L, N, V, alb, roughness, metallic, F0)
        Lo += np.where((shadow > 1e-3)[..., None], contrib, 0.0)
    if scene.get("prefiltered") is not None:                                                # :186-200
        ndv = np.maximum(dot(N, V), 1e-3)
        levels = scene["prefiltered"]
        spec, ok1 = cube_sample_lod(levels, Rv, roughness * len(levels))
        lut = texture_2d_repeat(scene["lut"], ndv, roughness)
        F = fresnel_schlick(ndv, F0)
        kD = (1.0 - F) * (1.0 - metallic)[:, None]
        spec = spec * (F * lut[..., 0:1] + lut[..., 1:2])
        irr, ok2 = cube_sample(scene["irradiance"], N)
        diffuse = kD * irr * alb
        Lo += (diffuse + spec) * ao[:, None]
        robust &= ok1 & ok2
    color = np.minimum(Lo + emissive, 10.0)                                                 # :202-203
    alpha = np.ones(n) if mat["alpha_mode"] == 1 else albedo[:, 3]                          # :205-208
    rgba = np.concatenate([color, alpha[:, None]], -1)
    rgba = np.where(masked[:, None], 0.0, rgba)                                             # :97: outColor = vec4(0)
    return rgba, robust, info


# ------------------------------------------------------------------ env.vert + env.frag


def env_frag(env_cube, inv_view_proj, xn, yn):
    """env.vert:22-26 (position = (invViewProj * (x, y, 1, 1)).xyz, no divide, interpolated over the quad: affine in
    NDC) + env.frag:10-17. xn, yn: NDC of the pixel centres. Returns (rgba, ok)."""
    pos = np.stack([xn, yn, np.ones_like(xn), np.ones_like(xn)], -1) @ inv_view_proj.T
    c, ok = cube_sample(env_cube, normalize(pos[..., :3]))
    c = c / (c + 1.0)
    c = c ** (1.0 / 2.2)
    return np.concatenate([c, np.ones(c.shape[:-1] + (1,))], -1), ok


# ------------------------------------------------------------------ envbake.vert + envbakediffuse.frag / envbakespecular.frag


def bake_face_inverse_view_proj(face):
    """VulkanRendererEnvironment.cpp:208-217, 224: proj = scale(1,-1,1), one view rotation per face, push inverse(proj*view)."""
    views = [rotate(np.radians(-90.0), 1), rotate(np.radians(90.0), 1), rotate(np.radians(90.0), 0), rotate(np.radians(-90.0), 0), np.eye(4),
             rotate(np.radians(180.0), 1)]
    proj = np.diag([1.0, -1.0, 1.0, 1.0])
    return np.linalg.inv(proj @ views[face])


def bake_positions(face, i, j, res):
    """envbake.vert: outPosition = (invviewproj * (x, y, 1, 1)).xyz at the centre of texel (i, j) of a res^2 target."""
    xn, yn = (i + 0.5) / res * 2.0 - 1.0, (j + 0.5) / res * 2.0 - 1.0
    inv = np.stack([bake_face_inverse_view_proj(f) for f in range(6)])[face]
    p = np.stack([xn, yn, np.ones_like(xn), np.ones_like(xn)], -1)
    return np.einsum("...rc,...c->...r", inv, p)[..., :3]


def bake_diffuse(env_cube, position):
    """envbakediffuse.frag:12-40 for positions [n, 3]. Returns (irradiance [n, 3], ok [n])."""
    normal = normalize(position)
    up = np.broadcast_to(np.array([0.0, 1.0, 0.0]), normal.shape)
    right = normalize(np.cross(up, normal))
    up = normalize(np.cross(normal, right))
    n_samples = 24
    step = 2 * PI / n_samples
    irr = np.zeros_like(normal)
    ok = np.ones(normal.shape[0], bool)
    for i in range(n_samples):
        for j in range(n_samples // 4):
            phi, theta = i * step, j * step
            ts = np.array([np.sin(theta) * np.cos(phi), np.sin(theta) * np.sin(phi), np.cos(theta)])
            vec = ts[0] * right + ts[1] * up + ts[2] * normal
            c, k = cube_sample(env_cube, vec)
            irr += c * np.cos(theta) * np.sin(theta)
            ok &= k
    return PI * irr * (1.0 / (n_samples * n_samples // 4)), ok


def radical_inverse_vdc(i):  # envbakespecular.frag:17-25
    bits = np.uint32(i)
    bits = ((bits << np.uint32(16)) | (bits >> np.uint32(16))) & np.uint32(0xFFFFFFFF)
    bits = ((bits & np.uint32(0x55555555)) << np.uint32(1)) | ((bits & np.uint32(0xAAAAAAAA)) >> np.uint32(1))
    bits = ((bits & np.uint32(0x33333333)) << np.uint32(2)) | ((bits & np.uint32(0xCCCCCCCC)) >> np.uint32(2))
    bits = ((bits & np.uint32(0x0F0F0F0F)) << np.uint32(4)) | ((bits & np.uint32(0xF0F0F0F0)) >> np.uint32(4))
    bits = ((bits & np.uint32(0x00FF00FF)) << np.uint32(8)) | ((bits & np.uint32(0xFF00FF00)) >> np.uint32(8))
    return float(bits) * 2.3283064365386963e-10


def importance_sample_ggx(xi, N, roughness):  # envbakespecular.frag:32-53
    a = roughness * roughness
    phi = 2.0 * PI * xi[0]
    cos_t = np.sqrt((1.0 - xi[1]) / (1.0 + (a * a - 1.0) * xi[1]))
    sin_t = np.sqrt(1.0 - cos_t * cos_t)
    H = np.array([np.cos(phi) * sin_t, np.sin(phi) * sin_t, cos_t])
    up = np.where((np.abs(N[..., 2]) < 0.999)[..., None], np.array([0.0, 0.0, 1.0]), np.array([1.0, 0.0, 0.0]))
    tangent = normalize(np.cross(up, N))
    bitangent = np.cross(N, tangent)
    return normalize(tangent * H[0] + bitangent * H[1] + N * H[2])


def bake_specular(env_cube, position, roughness):
    """envbakespecular.frag:55-86 for positions [n, 3]. Returns (colour [n, 3], ok [n])."""
    normal = normalize(position)
    V = normal
    n_samples = 128
    color = np.zeros_like(normal)
    total = np.zeros(normal.shape[0])
    ok = np.ones(normal.shape[0], bool)
    for i in range(n_samples):
        xi = (i / n_samples, radical_inverse_vdc(i))
        H = importance_sample_ggx(xi, normal, roughness)
        L = normalize(2.0 * dot(V, H)[..., None] * H - V)
        ndl = np.maximum(dot(normal, L), 0.0)
        c, k = cube_sample(env_cube, L)
        use = ndl > 0.0
        color += np.where(use[..., None], c * ndl[..., None], 0.0)
        total += np.where(use, ndl, 0.0)
        ok &= k | ~use
    return color / total[..., None], ok


# ------------------------------------------------------------------ rasterisation of a small scene, pixel centres only


def raster_triangles(clip, W, H, margin):
    """clip: [t, 3, 4] clip-space vertices of front-facing-or-not triangles, all with w > 0 (no clipping needed).
    For every pixel centre returns, per triangle: inside [t, H, W] (at least `margin` pixels inside every edge),
    unsure [t, H, W] (closer than `margin` to an edge or vertex), depth [t, H, W] and the perspective-correct
    barycentrics lam [t, H, W, 3]. Back faces (Vulkan: CCW front, negative area) are dropped (all False).
    Viewport (0, 0, W, H), depth range [0, 1] (VulkanRenderer.cpp:849)."""
    t = clip.shape[0]
    w = clip[..., 3]
    ndc = clip[..., :3] / w[..., None]
    X, Y = (ndc[..., 0] + 1.0) * 0.5 * W, (ndc[..., 1] + 1.0) * 0.5 * H
    px, py = np.meshgrid(np.arange(W) + 0.5, np.arange(H) + 0.5, indexing="xy")
    inside = np.zeros((t, H, W), bool)
    unsure = np.zeros((t, H, W), bool)
    depth = np.ones((t, H, W))
    lam = np.zeros((t, H, W, 3))
    for k in range(t):
        x, y = X[k], Y[k]
        # Vulkan: a = -1/2 sum(x_i y_i+1 - x_i+1 y_i); positive area = front facing under VK_FRONT_FACE_COUNTER_CLOCKWISE
        area = -0.5 * sum(x[i] * y[(i + 1) % 3] - x[(i + 1) % 3] * y[i] for i in range(3))
        if not area > 0:
            continue
        b = np.zeros((3, H, W))
        dist = np.zeros((3, H, W))
        for i in range(3):  # edge opposite vertex i, from vertex i+1 to vertex i+2
            x1, y1, x2, y2 = x[(i + 1) % 3], y[(i + 1) % 3], x[(i + 2) % 3], y[(i + 2) % 3]
            e = (x2 - x1) * (py - y1) - (y2 - y1) * (px - x1)
            b[i] = e / (-2.0 * area)
            dist[i] = e / (-np.hypot(x2 - x1, y2 - y1))  # signed distance to the edge in pixels, positive inside
        dmin = dist.min(0)
        inside[k] = dmin > margin
        unsure[k] = np.abs(dmin) <= margin
        depth[k] = b[0] * ndc[k, 0, 2] + b[1] * ndc[k, 1, 2] + b[2] * ndc[k, 2, 2]
        pw = np.stack([b[i] / w[k, i] for i in range(3)], -1)
        lam[k] = pw / pw.sum(-1, keepdims=True)
    return inside, unsure, depth, lam
