"""Known answers for the shading arithmetic: pbr.frag, env.frag and the two environment bakes, worked out in
numpy float64 from the GLSL (tests/glsl_f64.py, tests/kat_scene.py) and compared with what a back end renders.

The CPU tests pin the restatement in oracle/; the tests marked `gpu` put the HIP library through the same assertions,
so its shading is checked against the shader source directly and not only against the restatement.

Tolerances: the colour attachment is RGBA16F (relative step 2^-11, rounded to nearest), the cubes are B10G11R11
(6/6/5 mantissa bits, truncated), the back ends compute in float32.
"""
import numpy as np
import pytest

from renderer_amd.api import Renderer, default_settings
from tests import glsl_f64 as G
from tests import kat_scene as K

SETTINGS = dict(directional_shadow_map_resolution=512, point_shadow_map_resolution=64)


def _renderer(capi, W, H, msaa):
    return Renderer(capi, W, H, default_settings(msaa=msaa, **SETTINGS))


def _assert_close(got, want, mask, rtol, atol, what):
    err = np.abs(got - want) - (atol + rtol * np.abs(want))
    bad = (err > 0).any(-1) & mask
    if bad.any():
        y, x = np.argwhere(bad)[0]
        raise AssertionError(f"{what}: {bad.sum()} of {mask.sum()} pixels off, first at ({x},{y}): got {got[y, x]}, want {want[y, x]}")


def check_pbr(capi, msaa):
    W, H = 160, 96
    r = _renderer(capi, W, H, msaa)
    K.build(r, K.linear_environment(32))
    r.render_frame(exposure_mode=2, exposure=0.0)
    got = r.read_hdr().astype(np.float64)
    # with 4 samples a pixel must lie a whole pixel inside its triangle for all samples to see the same fragment
    want, valid = K.expected_hdr(r, W, H, margin=0.02 if msaa == 1 else 1.0)
    vis = r.read_visibility()
    n_frag = int((valid & (vis[0] != 0xFFFFFFFF)).sum())
    assert n_frag >= 2000, f"only {n_frag} shaded fragments are pinned"
    # 2^-11 from each of up to two RGBA16F roundings (opaque store, blend store), float32 arithmetic on top
    _assert_close(got, want, valid, rtol=1.5e-3, atol=2e-4, what=f"pbr.frag msaa {msaa}")
    return n_frag


def _segment_hits_box(P, light, prim, grow):
    """Does the segment from P [n, 3] to `light` pass through the box primitive `prim`, its faces moved outwards by
    `grow` (inwards when negative)? Slab test in the box's own coordinates."""
    inv = np.linalg.inv(prim.model)
    lo, hi = prim.geo.pos.min(0).astype(np.float64) - grow, prim.geo.pos.max(0).astype(np.float64) + grow
    a = np.concatenate([P, np.ones((P.shape[0], 1))], -1) @ inv.T
    b = np.append(np.asarray(light, np.float64), 1.0) @ inv.T
    o, d = a[:, :3], b[:3] - a[:, :3]
    with np.errstate(divide="ignore", invalid="ignore"):
        t0, t1 = (lo - o) / d, (hi - o) / d
    tmin, tmax = np.minimum(t0, t1).max(-1), np.maximum(t0, t1).min(-1)
    return (tmin <= tmax) & (tmax >= 0.0) & (tmin <= 1.0)


def check_detail(capi, msaa, variant="detail"):
    """Mip-mapped textures and shadowed point lights (the "detail" variant of tests/kat_scene.py), or the same scene
    with NEAREST filters in its Samplers (the "filters" variant)."""
    W, H = 240, 144
    r = _renderer(capi, W, H, msaa)
    built = K.build(r, K.linear_environment(32), variant=variant)
    r.render_frame(exposure_mode=2, exposure=0.0)
    got = r.read_hdr().astype(np.float64)
    det = {}
    want, valid = K.expected_hdr(r, W, H, margin=0.02 if msaa == 1 else 1.0, variant=variant, built=built, details=det)
    opaque = built[0]
    on_geometry = det["prim"] >= 0
    n_frag = int((valid & on_geometry).sum())
    assert n_frag >= 4000, f"only {n_frag} shaded fragments are pinned"
    # ---- what the pinned fragments cover
    lod = det["lod"]
    n_minified = int((valid & (lod >= 3.0)).sum())
    n_between = int((valid & (lod > 0.0) & (lod < 3.0)).sum())
    n_magnified = int((valid & (lod == 0.0)).sum())
    assert n_minified >= 500 and n_between >= 300 and n_magnified >= 300, (n_minified, n_between, n_magnified)
    assert np.nanmax(np.where(valid, lod, np.nan)) >= 5.0, "the far ground reaches the small levels of its 128^2 chain"
    sh = det["point_shadow"]  # [light, H, W]
    casting = [i for i, l in enumerate(K.DETAIL_LIGHTS) if l[2]]
    partly = np.zeros((H, W), bool)
    fully = np.zeros((H, W), bool)
    for i in casting:
        partly |= (sh[i] > 1e-3) & (sh[i] < 1.0 - 1e-3)
        fully |= sh[i] == 0.0
    n_partly, n_fully = int((valid & partly).sum()), int((valid & fully).sum())
    assert n_partly >= 100 and n_fully >= 500, (n_partly, n_fully)
    # ---- where the cube faces put the shadows, against the geometry of the scene (not against any depth map): ground
    # fragments whose line of sight to the light passes well inside one of the two boxes are in full shadow; those
    # under the light (its -Y face, where the ground's depth is nearly constant over a texel) that clear every
    # occluder by a wide margin are fully lit
    ground = det["prim"] == 0
    box, wall, mover = opaque[1], opaque[2], opaque[3]
    ys, xs = np.nonzero(ground & valid)
    P = det["world"][ys, xs]
    for i in casting:
        light = np.asarray(K.DETAIL_LIGHTS[i][0], np.float64)
        s_i = sh[i][ys, xs]
        umbra = _segment_hits_box(P, light, box, -0.2) | _segment_hits_box(P, light, mover, -0.2)
        assert umbra.sum() >= 50 and np.all(s_i[umbra & ~np.isnan(s_i)] == 0.0), f"light {i}: umbra of the boxes"
        wall_box = type("W", (), {})()
        wall_box.model, wall_box.geo = wall.model, wall.geo
        clear = ~(_segment_hits_box(P, light, box, 0.4) | _segment_hits_box(P, light, mover, 0.4) | _segment_hits_box(P, light, wall_box, 0.4))
        L = light - P
        under = np.abs(L[:, 1]) > 1.5 * np.maximum(np.abs(L[:, 0]), np.abs(L[:, 2]))
        sel = clear & under & ~np.isnan(s_i)
        if i == casting[0]:
            assert sel.sum() >= 200
        assert np.all(s_i[sel] == 1.0), f"light {i}: open ground under the light"
    # ---- and the colours: 2^-11 from each of up to two RGBA16F roundings, float32 arithmetic on top
    _assert_close(got, want, valid, rtol=1.5e-3, atol=2e-4, what=f"pbr.frag, {variant} scene, msaa {msaa}")
    per_prim = [int((valid & (det["prim"] == k)).sum()) for k in range(len(opaque))]
    if variant == "filters":
        # every filter combination of the scene keeps enough pinned fragments away from its texel and level boundaries
        assert per_prim[0] >= 1500 and per_prim[1] >= 150 and per_prim[2] >= 100 and per_prim[3] >= 40, per_prim
    return dict(fragments=n_frag, minified=n_minified, magnified=n_magnified, partly_shadowed=n_partly, fully_shadowed=n_fully, per_prim=per_prim)


def check_sky(capi):
    W, H = 64, 40
    r = _renderer(capi, W, H, 1)
    K.build(r, K.linear_environment(16), with_meshes=False)
    r.render_frame(exposure_mode=2, exposure=0.0)
    got = r.read_hdr().astype(np.float64)
    want, ok = K.expected_sky(r, W, H)
    assert ok.sum() >= W * H - 64
    _assert_close(got, want, ok, rtol=1e-3, atol=1e-5, what="env.frag")
    # closed form for a constant environment: c / (c + 1), then ^(1 / 2.2); exact inputs (0.5, 1, 0.25)
    r2 = _renderer(capi, 8, 8, 1)
    K.build(r2, K.constant_environment(4, (0.5, 1.0, 0.25)), with_meshes=False)
    r2.render_frame(exposure_mode=2, exposure=0.0)
    want2 = np.array([(1 / 3) ** (1 / 2.2), 0.5 ** (1 / 2.2), 0.2 ** (1 / 2.2), 1.0])
    assert np.allclose(r2.read_hdr(), want2, rtol=1e-3)


def _packed_bounds(want, tol):
    """What a B10G11R11 texel may hold for a float64 value `want` computed in float32 and truncated on store."""
    step = np.array([2.0 ** -6, 2.0 ** -6, 2.0 ** -5])
    return want * (1.0 - tol) * (1.0 - step) - 1e-7, want * (1.0 + tol)


def check_bakes(capi):
    r = _renderer(capi, 16, 16, 1)
    faces = K.linear_environment(32)
    K.build(r, faces, with_meshes=False)
    env = r.read_env_cube(0).astype(np.float64)
    # the stored environment is the input truncated to the packed format
    lo, hi = _packed_bounds(faces[..., :3].astype(np.float64), 0.0)
    assert np.all(env >= lo) and np.all(env <= hi)
    rng = np.random.default_rng(17)
    # ---- envbakediffuse.frag at 256^2: 200 texels over the six faces, and the texels next to face edges
    irr = r.read_env_cube(1).astype(np.float64)
    assert irr.shape == (6, 256, 256, 3)
    face = rng.integers(0, 6, 200)
    i, j = rng.integers(0, 256, 200), rng.integers(0, 256, 200)
    i[:12], j[12:24] = 0, 255
    want, ok = G.bake_diffuse(env, G.bake_positions(face, i, j, 256))
    assert ok.sum() >= 150
    lo, hi = _packed_bounds(want, 2e-3)
    got = irr[face, j, i]
    assert np.all((got >= lo)[ok]) and np.all((got <= hi)[ok]), "irradiance bake"
    # ---- envbakespecular.frag, 10 levels of 512 >> l with roughness l / 9
    n_checked = 0
    for level in range(10):
        res = 512 >> level
        pre = r.read_env_cube(2, level).astype(np.float64)
        assert pre.shape == (6, res, res, 3)
        n = 40
        face, i, j = rng.integers(0, 6, n), rng.integers(0, res, n), rng.integers(0, res, n)
        want, ok = G.bake_specular(env, G.bake_positions(face, i, j, res), level / 9.0)
        lo, hi = _packed_bounds(want, 2e-3)
        got = pre[face, j, i]
        assert np.all((got >= lo)[ok]) and np.all((got <= hi)[ok]), f"prefilter level {level}"
        n_checked += int(ok.sum())
    assert n_checked >= 300
    # ---- constant environment: closed forms. Irradiance = pi * c * sum_{j<6} cos(15 j) sin(15 j) / 6 (24 x 6 samples,
    # normalised by 144); every prefiltered level = c (a weighted mean of a constant)
    c = np.array([0.5, 1.0, 0.25])
    r2 = _renderer(capi, 16, 16, 1)
    K.build(r2, K.constant_environment(8, c), with_meshes=False)
    th = np.radians(15.0) * np.arange(6)
    want = np.pi * c * np.sum(np.cos(th) * np.sin(th)) / 6.0
    lo, hi = _packed_bounds(want, 1e-5)
    irr = r2.read_env_cube(1).astype(np.float64)
    assert np.all(irr >= lo) and np.all(irr <= hi)
    for level in range(10):
        pre = r2.read_env_cube(2, level).astype(np.float64)
        lo, hi = _packed_bounds(c, 1e-5)
        assert np.all(pre >= lo) and np.all(pre <= hi), level


@pytest.mark.parametrize("msaa", [1, 4])
def test_pbr_frag_known_answers(orc_capi, msaa):
    check_pbr(orc_capi, msaa)


def test_env_frag_known_answers(orc_capi):
    check_sky(orc_capi)


def test_environment_bakes_known_answers(orc_capi):
    check_bakes(orc_capi)


@pytest.mark.parametrize("msaa", [1, 4])
def test_mip_mapped_textures_and_shadowed_point_lights(orc_capi, msaa):
    print(check_detail(orc_capi, msaa))


def check_jittered(capi, seed, monkeypatch):
    """The textured scene from another camera, under other lights, through another variant of its materials: the same
    float64 expectations, so that the known answers do not hang on one view of one arrangement."""
    rng = np.random.default_rng(seed)
    cam = dict(K.CAMERA)
    cam["position"] = tuple(np.asarray(cam["position"]) + rng.uniform(-0.6, 0.6, 3))
    cam["euler"] = tuple(np.asarray(cam["euler"]) + np.array([rng.uniform(-0.08, 0.08), rng.uniform(-0.15, 0.15), 0.0]))
    cam["fov"] = float(np.float32(np.radians(rng.uniform(32.0, 52.0))))
    lights = [(tuple(np.asarray(p) + rng.uniform(-0.7, 0.7, 3)), tuple(np.asarray(i) * rng.uniform(0.5, 1.5, 3)), cast) for p, i, cast in K.DETAIL_LIGHTS]
    plain = [(tuple(np.asarray(p) + rng.uniform(-0.7, 0.7, 3)), tuple(np.asarray(i) * rng.uniform(0.5, 1.5, 3))) for p, i in K.POINT_LIGHTS]
    if 2000 <= seed < 3000:
        # lights a few centimetres above the ground (1 / d^2 of a very small d^2, light vectors that turn by a radian over a
        # few pixels); the two that cast shadows stay half a metre up (nothing may come nearer to them than the 0.1 of
        # their cube's near plane)
        above = lambda h0, h1: (lambda x: (x, 0.1003 * x + float(rng.uniform(h0, h1)), float(rng.uniform(-7.0, 1.5))))(float(rng.uniform(-3.5, 3.5)))  # noqa: E731
        lights = [(above(0.5, 1.0) if cast else above(0.03, 0.3), tuple(np.asarray(i) * (rng.uniform(0.05, 0.2) if cast else rng.uniform(0.002, 0.02))), cast)
                  for p, i, cast in K.DETAIL_LIGHTS]
        plain = [(above(0.03, 0.3), tuple(np.asarray(i) * rng.uniform(0.002, 0.02))) for p, i in K.POINT_LIGHTS]
    if 3000 <= seed < 4000:
        # the eye a hand above the ground, looking along it: N.V near 0 on everything horizontal (the 1e-3 floors of the
        # geometry term, the edge of the BRDF table, Fresnel near 1), the far ground on the last levels of its mip chains
        x = float(rng.uniform(-2.5, 2.5))
        cam["position"] = (x, 0.1003 * x + float(rng.uniform(0.12, 0.3)), float(rng.uniform(3.9, 4.5)))   # in front of the ground's near edge (z = 3)
        cam["euler"] = (float(rng.uniform(-0.03, 0.02)), float(rng.uniform(-0.04, 0.04)), 0.0)   # (the ground's corners stay in front of the eye)
    if 1000 <= seed < 2000:
        # against the light: the camera low over the ground, the lights low on the far side of the scene, so that the ground
        # between them is seen at grazing angles with V.L near -1 (forward highlights, where N.H is formed from nearly
        # opposite vectors)
        cam["position"] = (float(rng.uniform(-2.0, 2.0)), float(rng.uniform(0.35, 1.0)), float(rng.uniform(6.0, 8.0)))
        cam["euler"] = (float(rng.uniform(-0.12, -0.02)), float(rng.uniform(-0.15, 0.15)), 0.0)
        far_side = lambda: (float(rng.uniform(-4.0, 4.0)), float(rng.uniform(0.2, 1.2)), float(rng.uniform(-22.0, -6.0)))  # noqa: E731
        lights = [(far_side(), tuple(np.asarray(i) * rng.uniform(2.0, 8.0)), cast) for p, i, cast in K.DETAIL_LIGHTS]
        plain = [(far_side(), tuple(np.asarray(i) * rng.uniform(2.0, 8.0))) for p, i in K.POINT_LIGHTS]
        original_prims = K.prims

        def glossy_prims(variant="base"):   # roughness factors quartered: highlights a few pixels wide
            opaque, blended = original_prims(variant)
            for p in opaque + blended:
                p.mat["roughness"] = float(np.float32(p.mat["roughness"] * 0.25))
            return opaque, blended
        monkeypatch.setattr(K, "prims", glossy_prims)
    monkeypatch.setattr(K, "CAMERA", cam)
    monkeypatch.setattr(K, "DETAIL_LIGHTS", lights)
    monkeypatch.setattr(K, "POINT_LIGHTS", plain)
    variant = ["detail", "filters", "float", "base"][seed % 4]   # "base": 1x1 textures, five lights without maps, MASK and BLEND quads
    msaa = [1, 4][(seed // 4) % 2]
    W, H = int(rng.integers(150, 260)), int(rng.integers(90, 150))
    r = _renderer(capi, W, H, msaa)
    built = K.build(r, K.linear_environment(32), variant=variant)
    r.render_frame(exposure_mode=2, exposure=0.0)
    got = r.read_hdr().astype(np.float64)
    det = {}
    want, valid = K.expected_hdr(r, W, H, margin=0.02 if msaa == 1 else 1.0, variant=variant, built=built, details=det)
    n_frag = int((valid & (det["prim"] >= 0)).sum())
    assert n_frag >= (800 if seed >= 1000 else 1200), f"only {n_frag} shaded fragments are pinned"
    # (against the light with glossy materials the float32 evaluation of a narrow lobe is itself worth 2e-3 on single pixels)
    _assert_close(got, want, valid, rtol=2.5e-3 if 1000 <= seed < 2000 else 1.5e-3, atol=2e-4, what=f"pbr.frag, {variant} scene, seed {seed}, msaa {msaa}, {W}x{H}")
    r.close()
    return n_frag


@pytest.mark.parametrize("seed", [1, 2, 3, 4])
def test_other_views_of_the_textured_scene(orc_capi, seed, monkeypatch):
    print(check_jittered(orc_capi, seed, monkeypatch))


@pytest.mark.parametrize("seed", [1000, 1001])
def test_grazing_highlights_against_the_light(orc_capi, seed, monkeypatch):
    """Camera low over the ground, lights low on the far side, glossy materials: V.L near -1, where the length of the half
    vector must come from the sum of the vectors and not from 1 + 2 V.L + L.L."""
    print(check_jittered(orc_capi, seed, monkeypatch))


@pytest.mark.gpu
@pytest.mark.parametrize("seed", list(range(1000, 1012)))
def test_grazing_highlights_against_the_light_hip(hip_capi, seed, monkeypatch):
    print(check_jittered(hip_capi, seed, monkeypatch))


@pytest.mark.parametrize("seed", [2000, 3001])
def test_lights_close_to_the_ground_and_grazing_views(orc_capi, seed, monkeypatch):
    """Seeds from 2000: point lights a few centimetres above the ground. Seeds from 3000: the eye a hand above the ground,
    looking along it (N.V near 0, the last mip levels)."""
    print(check_jittered(orc_capi, seed, monkeypatch))


@pytest.mark.gpu
@pytest.mark.parametrize("seed", list(range(2000, 2008)) + list(range(3000, 3008)))
def test_lights_close_to_the_ground_and_grazing_views_hip(hip_capi, seed, monkeypatch):
    print(check_jittered(hip_capi, seed, monkeypatch))


@pytest.mark.gpu
@pytest.mark.parametrize("seed", list(range(1, 13)))
def test_other_views_of_the_textured_scene_hip(hip_capi, seed, monkeypatch):
    print(check_jittered(hip_capi, seed, monkeypatch))


@pytest.mark.parametrize("msaa", [1, 4])
def test_nearest_sampler_filters(orc_capi, msaa):
    """The Sampler's magFilter / minFilter / mipmapFilter (VulkanRenderer.cpp:363-385) by the Vulkan rules, in float64."""
    print(check_detail(orc_capi, msaa, variant="filters"))


def test_float_image_formats(orc_capi):
    """The SFLOAT formats of Image.h:9-12 as material textures: float texels filtered as they are, 16-bit ones rounded
    level by level, a missing alpha read as 1; the same scene, the same float64 expectations."""
    print(check_detail(orc_capi, 1, variant="float"))


@pytest.mark.gpu
@pytest.mark.parametrize("msaa", [1, 4])
def test_float_image_formats_hip(hip_capi, msaa):
    print(check_detail(hip_capi, msaa, variant="float"))


@pytest.mark.gpu
@pytest.mark.parametrize("msaa", [1, 4])
def test_nearest_sampler_filters_hip(hip_capi, msaa):
    print(check_detail(hip_capi, msaa, variant="filters"))


@pytest.mark.gpu
@pytest.mark.parametrize("msaa", [1, 4])
def test_pbr_frag_known_answers_hip(hip_capi, msaa):
    check_pbr(hip_capi, msaa)


@pytest.mark.gpu
@pytest.mark.parametrize("msaa", [1, 4])
def test_mip_mapped_textures_and_shadowed_point_lights_hip(hip_capi, msaa):
    print(check_detail(hip_capi, msaa))


@pytest.mark.gpu
def test_env_frag_known_answers_hip(hip_capi):
    check_sky(hip_capi)


@pytest.mark.gpu
def test_environment_bakes_known_answers_hip(hip_capi):
    check_bakes(hip_capi)
