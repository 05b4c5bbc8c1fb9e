"""GPU parity tests: the HIP library (through the C ABI) against the CPU restatement on the same inputs.

Bars: depth, coverage/visibility, shadow maps, mip chains, IBL cubes and the bloom chain are compared
bit for bit; the tonemapped 8-bit frame must be within 1/255 per channel and >= 45 dB PSNR.
"""
import numpy as np
import pytest

from renderer_amd import scenes
from tests import helpers as hp

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("size", [(64, 48), (250, 141), (640, 360), (391, 287), (128, 135), (138, 26), (20, 300)])
def test_post_process_parity(orc_capi, hip_capi, size):
    W, H = size
    ro, rh = hp.make_pair(orc_capi, hip_capi, W, H)
    hdr = hp.random_hdr(W, H, seed=W)
    out_o, avg_o = ro.post_process(hdr, 1.5)
    out_h, avg_h = rh.post_process(hdr, 1.5)
    for level in range(6):
        bo, bh = ro.read_bloom(level), rh.read_bloom(level)
        assert np.array_equal(bo, bh), f"bloom mip {level} differs"
    hp.assert_image_close(out_o, out_h)
    lo, lh = ro.read_log_luminance(), rh.read_log_luminance()
    fin = np.isfinite(lo)
    assert np.array_equal(fin, np.isfinite(lh))
    assert np.allclose(lo[fin], lh[fin], atol=1e-5)
    if np.isfinite(avg_o):
        assert abs(avg_o - avg_h) < 1e-4
    else:
        assert avg_o == avg_h


@pytest.mark.parametrize("msaa", [1, 4])
def test_small_scene_parity(orc_capi, hip_capi, msaa):
    W, H = 320, 180
    ro, rh = hp.make_pair(orc_capi, hip_capi, W, H, **hp.small_settings(msaa))
    scenes.small_scene(ro)
    scenes.small_scene(rh)
    out_o, avg_o = ro.render_frame(exposure_mode=1, temporal_luminance=0.5)
    out_h, avg_h = rh.render_frame(exposure_mode=1, temporal_luminance=0.5)
    # host logic: cascade matrices bit-identical
    vo, do = ro.read_cascades()
    vh, dh = rh.read_cascades()
    assert np.array_equal(vo, vh) and np.array_equal(do, dh)
    # rasteriser: bit-exact depth and coverage
    for i in range(4):
        assert np.array_equal(ro.read_directional_shadow(i), rh.read_directional_shadow(i)), f"cascade {i}"
    for slot in range(2):
        for face in range(6):
            assert np.array_equal(ro.read_point_shadow(slot, face), rh.read_point_shadow(slot, face)), f"point {slot}/{face}"
    assert np.array_equal(ro.read_visibility(), rh.read_visibility())
    assert np.array_equal(ro.read_depth(), rh.read_depth())
    so, sh = ro.frame_stats(), rh.frame_stats()
    assert (so.triangle_instances, so.triangles_binned, so.views) == (sh.triangle_instances, sh.triangles_binned, sh.views)
    # shading: tolerance on the tonemapped output
    hp.assert_image_close(out_o, out_h)
    assert abs(avg_o - avg_h) < 1e-3


@pytest.mark.parametrize("msaa", [1, 4])
def test_many_blended_layers(orc_capi, hip_capi, msaa):
    """Seven screen-filling translucent and alpha-tested layers in front of a floor, some of them huge triangles that
    are clipped by the near plane and the guard band: every 8x8 quadrant collects several hundred blended fragments
    (its list space is extended many times), blend order is the back-to-front draw order, and the layers exercise
    the clipped-polygon path of the blended pass next to the 32-bit edge functions of the small triangles."""
    W, H = 200, 120
    ro, rh = hp.make_pair(orc_capi, hip_capi, W, H, **hp.small_settings(msaa))
    for r in (ro, rh):
        tex = scenes.material_textures(r, 32, 91, tint=(0.9, 0.9, 0.9), alpha=0.25 + 0.5 * scenes.value_noise(32, 5, octaves=2, base=4))
        r.add_mesh([scenes.upload_primitive(r, scenes.plane(20, 20, 4, 4, uv_scale=4.0), scenes.make_material(tex))])
        for k in range(7):
            big = k % 3 == 2   # one quad of two huge triangles instead of a fine grid
            quad = scenes.plane(400.0 if big else 9.0, 400.0 if big else 6.0, 1 if big else 12, 1 if big else 8, uv_scale=60.0 if big else 2.0)
            quad = quad.transformed(scenes.translation(0.1 * k, 1.2, 2.5 - 0.45 * k) @ scenes.rotation_x(np.pi / 2 + 0.05 * k))
            mode = scenes.B.ALPHA_MASK if k % 4 == 1 else scenes.B.ALPHA_BLEND
            mat = scenes.make_material(tex, base_color=(0.4 + 0.08 * k, 1.0 - 0.1 * k, 0.6, 0.35 + 0.05 * k), alpha_mode=mode, alpha_cutoff=0.45,
                                       metallic=0.2, roughness=0.6)
            r.add_mesh([scenes.upload_primitive(r, quad, mat, scenes.B.UINT16)])
        r.set_brdf_lut(scenes.brdf_lut(32, 32))
        r.set_environment_map(scenes.sky_faces(16))
        r.set_camera(position=(0.0, 1.3, 4.6), euler=(-0.05, 0.0, 0.0))
        r.set_lights([((1.5, 2.2, 3.5), (30.0, 24.0, 18.0), True)], scenes.default_directional_light())
    out_o, _ = ro.render_frame(exposure_mode=1, temporal_luminance=0.5)
    out_h, _ = rh.render_frame(exposure_mode=1, temporal_luminance=0.5)
    assert np.array_equal(ro.read_depth(), rh.read_depth())
    hdr_o, hdr_h = ro.read_hdr(), rh.read_hdr()
    assert np.allclose(hdr_o[..., 3], hdr_h[..., 3], rtol=2e-3, atol=1e-4), "alpha is the last blended fragment's (a filtered texel: 1-2 half ulps)"
    hp.assert_image_close(out_o, out_h)
    assert hp.psnr(out_o, out_h) > 45
    # the layers are really there: the frame differs from the same scene without them
    assert (np.abs(hdr_o[..., 3] - 1.0) > 1e-3).mean() > 0.5


@pytest.mark.parametrize("msaa", [1, 4])
def test_degenerate_scenes(orc_capi, hip_capi, msaa):
    """What a caller can legitimately leave out: no mesh at all (sky only), no lights set, no environment and no
    BRDF table (black sky, no image-based term), and a mesh entirely behind the camera. Sizes that are not multiples
    of the 16-pixel tile."""
    W, H = 173, 99
    frame = dict(exposure_mode=1, temporal_luminance=0.5)

    def pair():
        return hp.make_pair(orc_capi, hip_capi, W, H, **hp.small_settings(msaa))

    # 1: nothing but the environment
    ro, rh = pair()
    for r in (ro, rh):
        r.set_environment_map(scenes.sky_faces(16))
        r.set_brdf_lut(scenes.brdf_lut(16, 16))
        r.set_camera(position=(0, 1, 3), euler=(0.2, 0.4, 0.0))
    a, la = ro.render_frame(**frame)
    b, lb = rh.render_frame(**frame)
    assert (ro.read_depth() == 1.0).all() and np.array_equal(ro.read_depth(), rh.read_depth())
    assert (rh.read_visibility() == 0xFFFFFFFF).all()
    hp.assert_image_close(a, b)
    assert a[..., :3].std() > 2, "the sky is visible"
    ro.close(); rh.close()

    # 2: geometry, but neither lights nor environment nor table: emissive and nothing else
    ro, rh = pair()
    for r in (ro, rh):
        tex = scenes.material_textures(r, 16, 7)
        r.add_mesh([scenes.upload_primitive(r, scenes.box(1.0, 1.0, 1.0, 2), scenes.make_material(tex, emissive=(0.5, 0.4, 0.3)), scenes.B.UINT16)])
        r.set_camera(position=(1.2, 1.0, 2.5), euler=(-0.3, 0.45, 0.0))
    a, _ = ro.render_frame(**frame)
    b, _ = rh.render_frame(**frame)
    assert np.array_equal(ro.read_depth(), rh.read_depth()) and np.array_equal(ro.read_visibility(), rh.read_visibility())
    hp.assert_image_close(a, b)
    assert (ro.read_depth() < 1.0).any()
    # 3: the same box behind the camera: every triangle is clipped away or culled
    for r in (ro, rh):
        r.set_camera(position=(0.0, 0.0, -3.0), euler=(0.0, 0.0, 0.0))
    a, _ = ro.render_frame(**frame)
    b, _ = rh.render_frame(**frame)
    assert (rh.read_depth() == 1.0).all() and np.array_equal(a, b)
    so, sh = ro.frame_stats(), rh.frame_stats()
    assert so.triangles_binned == sh.triangles_binned == 0
    ro.close(); rh.close()


@pytest.mark.parametrize("msaa", [1, 4])
def test_chunk_culling_changes_nothing(orc_capi, hip_capi, msaa):
    """A finely tessellated floor far larger than the view, seen from close above it through a small frame whose size
    is not a multiple of the tile: most 64-triangle chunks lie beside or behind the frustum and the GPU pre-pass culls
    them (the statistic says so), some straddle the near plane and the sides of the viewport. The CPU checker has no
    such pre-pass; depth, visibility, the cascades and the number of triangles that reach the tile lists must not
    notice."""
    W, H = 71, 53
    ro, rh = hp.make_pair(orc_capi, hip_capi, W, H, **hp.small_settings(msaa))
    for r in (ro, rh):
        tex = scenes.material_textures(r, 16, 3)
        floor = scenes.plane(80.0, 80.0, 96, 96, uv_scale=16.0)
        r.add_mesh([scenes.upload_primitive(r, floor, scenes.make_material(tex))])
        r.add_mesh([scenes.upload_primitive(r, scenes.box(1.0, 2.0, 1.0, 6).transformed(scenes.translation(0.5, 1.0, -1.0)),
                                            scenes.make_material(tex), scenes.B.UINT16)])
        r.set_brdf_lut(scenes.brdf_lut(16, 16))
        r.set_environment_map(scenes.sky_faces(16))
        r.set_lights([((2.0, 3.0, 1.0), (30.0, 30.0, 30.0), True)], scenes.default_directional_light())
    for position, euler in (((0.3, 0.6, 2.5), (-0.15, 0.2, 0.0)), ((-30.0, 0.4, 35.0), (-0.05, -0.7, 0.0)), ((0.0, 25.0, 0.0), (-1.5, 0.0, 0.0))):
        for r in (ro, rh):
            r.set_camera(position=position, euler=euler)
        out_o, _ = ro.render_frame(exposure_mode=1, temporal_luminance=0.5)
        out_h, _ = rh.render_frame(exposure_mode=1, temporal_luminance=0.5)
        assert np.array_equal(ro.read_depth(), rh.read_depth())
        assert np.array_equal(ro.read_visibility(), rh.read_visibility())
        for i in range(4):
            assert np.array_equal(ro.read_directional_shadow(i), rh.read_directional_shadow(i)), f"cascade {i}"
        for face in range(6):
            assert np.array_equal(ro.read_point_shadow(0, face), rh.read_point_shadow(0, face)), f"point face {face}"
        so, sh = ro.frame_stats(), rh.frame_stats()
        assert (so.triangle_instances, so.triangles_binned) == (sh.triangle_instances, sh.triangles_binned)
        assert so.chunks_culled == 0 and sh.chunks_culled > 0
        hp.assert_image_close(out_o, out_h)
    assert (rh.read_depth() < 1.0).any()


@pytest.mark.parametrize("size", [(64, 64), (32, 8), (1, 1), (37, 21), (100, 3), (5, 250), (257, 129)])
def test_mip_chain_parity(orc_capi, hip_capi, size):
    """loadImage builds the chain by successive LINEAR blits (VulkanRenderer.cpp:311-346): every level of an sRGB and
    of a UNORM image, square and ragged sizes (odd sizes take the general blit weights, not the 2x2 box), three-channel
    input included, must come out bit for bit as the CPU restatement stores them."""
    w, h = size
    ro, rh = hp.make_pair(orc_capi, hip_capi, 16, 16)
    rng = np.random.default_rng(w * 1000 + h)
    n_levels = int(np.floor(np.log2(max(w, h)))) + 1
    for fmt, ch in ((scenes.B.FORMAT_R8G8B8A8_SRGB, 4), (scenes.B.FORMAT_R8G8B8A8_UNORM, 4), (scenes.B.FORMAT_R8G8B8_SRGB, 3),
                    (scenes.B.FORMAT_R8G8B8_UNORM, 3)):
        img = rng.integers(0, 256, size=(h, w, ch), dtype=np.uint8)
        io, ih = ro.load_image(img, fmt), rh.load_image(img, fmt)
        for level in range(n_levels):
            lo, lh = ro.read_image_level(io, level), rh.read_image_level(ih, level)
            assert lo.shape == (max(1, h >> level), max(1, w >> level), 4)
            assert np.array_equal(lo, lh), f"format {fmt} level {level} of {w}x{h}"
        with pytest.raises(Exception):
            rh.read_image_level(ih, n_levels)
    # the SFLOAT formats (Image.h:9-12): float texels, the 16-bit ones rounded to binary16 level by level
    for fmt, ch, dt in ((scenes.B.FORMAT_R32G32B32A32_SFLOAT, 4, np.float32), (scenes.B.FORMAT_R32G32B32_SFLOAT, 3, np.float32),
                        (scenes.B.FORMAT_R16G16B16A16_SFLOAT, 4, np.float16), (scenes.B.FORMAT_R16G16B16_SFLOAT, 3, np.float16)):
        img = (rng.random((h, w, ch), dtype=np.float32) ** 3 * 8.0).astype(dt)
        io, ih = ro.load_image(img, fmt), rh.load_image(img, fmt)
        for level in range(n_levels):
            lo, lh = ro.read_image_level(io, level), rh.read_image_level(ih, level)
            assert lo.dtype == np.float32 and lo.shape == (max(1, h >> level), max(1, w >> level), 4)
            assert np.array_equal(lo.view(np.uint32), lh.view(np.uint32)), f"format {fmt} level {level} of {w}x{h}"
    # max_mip_levels cuts the chain short
    img = rng.integers(0, 256, size=(h, w, 4), dtype=np.uint8)
    io, ih = ro.load_image(img, scenes.B.FORMAT_R8G8B8A8_UNORM, max_mip_levels=2), rh.load_image(img, scenes.B.FORMAT_R8G8B8A8_UNORM, max_mip_levels=2)
    for level in range(min(2, n_levels)):
        assert np.array_equal(ro.read_image_level(io, level), rh.read_image_level(ih, level))
    with pytest.raises(Exception):
        rh.read_image_level(ih, min(2, n_levels))


def test_exposure_recurrence_parity(orc_capi, hip_capi):
    """exposure_mode 0 (VulkanRenderer.cpp:1075-1082): the exposure of a frame follows from the average log-luminance
    of the frame before it and the time step. Six frames with a moving camera and uneven time steps on both back
    ends: the exposure pushed to the tonemap pass, the luminance fed back and the 8-bit frames must agree."""
    W, H = 320, 280  # the luminance reduction reads a 256x256 window
    ro, rh = hp.make_pair(orc_capi, hip_capi, W, H, **hp.small_settings(4))
    scenes.small_scene(ro)
    scenes.small_scene(rh)
    steps = [1 / 60, 1 / 30, 0.5, 1 / 60, 2.0, 0.1]
    exposures = []
    for k, dt in enumerate(steps):
        for r in (ro, rh):
            r.set_camera(position=(0.4 + 0.25 * k, 1.7, 5.2 - 0.3 * k), euler=(-0.2, 0.05 + 0.04 * k, 0.0))
        out_o, avg_o = ro.render_frame(exposure_mode=0, delta_time=dt)
        out_h, avg_h = rh.render_frame(exposure_mode=0, delta_time=dt)
        eo, eh = ro.frame_stats().exposure, rh.frame_stats().exposure
        assert abs(eo - eh) <= 2e-4 * abs(eo), f"frame {k}: exposure {eo} vs {eh}"
        assert abs(avg_o - avg_h) < 1e-3, f"frame {k}"
        hp.assert_image_close(out_o, out_h)
        exposures.append(eo)
    # first frame: the read-back holds 0, so the adapted luminance moves from 0 towards 2^0 by 1 - e^-dt; later frames adapt
    tl = 1.0 - np.exp(-steps[0])
    assert abs(exposures[0] - 0.18 / (1.03 - 2.0 / (np.log2(tl + 1.0) + 2.0))) < 1e-3
    assert len({round(e, 4) for e in exposures}) == len(exposures)


@pytest.mark.parametrize("msaa", [1, 4])
def test_blended_pass_in_bounded_memory(orc_capi, hip_capi, msaa, monkeypatch):
    """Twelve screen-filling translucent layers with a fragment-list budget far below what they need at once
    (RDR_BLEND_BUDGET_SLOTS, read when the context is created): the library counts the slots each tile takes in a dry
    run and renders the tiles in groups. The frame must equal, byte for byte, the one rendered with the default budget
    (all tiles at once), and agree with the CPU checker, which blends in constant memory as the reference does."""
    W, H = 200, 120

    def build(r):
        tex = scenes.material_textures(r, 32, 91, tint=(0.9, 0.9, 0.9), alpha=0.25 + 0.5 * scenes.value_noise(32, 5, octaves=2, base=4))
        r.add_mesh([scenes.upload_primitive(r, scenes.plane(20, 20, 4, 4, uv_scale=4.0), scenes.make_material(tex))])
        for k in range(12):
            quad = scenes.plane(400.0, 400.0, 1, 1, uv_scale=60.0) if k % 3 == 2 else scenes.plane(9.0, 6.0, 12, 8, uv_scale=2.0)
            quad = quad.transformed(scenes.translation(0.05 * k, 1.2, 2.8 - 0.3 * k) @ scenes.rotation_x(np.pi / 2 + 0.03 * k))
            mode = scenes.B.ALPHA_MASK if k % 4 == 1 else scenes.B.ALPHA_BLEND
            mat = scenes.make_material(tex, base_color=(0.4 + 0.04 * k, 1.0 - 0.06 * k, 0.6, 0.3 + 0.04 * k), alpha_mode=mode, alpha_cutoff=0.45,
                                       metallic=0.2, roughness=0.6)
            r.add_mesh([scenes.upload_primitive(r, quad, mat, scenes.B.UINT16)])
        r.set_brdf_lut(scenes.brdf_lut(32, 32))
        r.set_environment_map(scenes.sky_faces(16))
        r.set_camera(position=(0.0, 1.3, 4.6), euler=(-0.05, 0.0, 0.0))
        r.set_lights([((1.5, 2.2, 3.5), (30.0, 24.0, 18.0), True)], scenes.default_directional_light())

    frame = dict(exposure_mode=1, temporal_luminance=0.5)
    ro, r_all = hp.make_pair(orc_capi, hip_capi, W, H, **hp.small_settings(msaa))
    monkeypatch.setenv("RDR_BLEND_BUDGET_SLOTS", "20000")
    _, r_groups = hp.make_pair(orc_capi, hip_capi, W, H, **hp.small_settings(msaa))
    monkeypatch.delenv("RDR_BLEND_BUDGET_SLOTS")
    for r in (ro, r_all, r_groups):
        build(r)
    out_o, _ = ro.render_frame(**frame)
    out_a, _ = r_all.render_frame(**frame)
    out_g, _ = r_groups.render_frame(**frame)
    assert r_all.frame_stats().blend_groups == 0 and r_groups.frame_stats().blend_groups >= 4
    assert np.array_equal(out_a, out_g), "rendering the blended tiles in groups must not change the frame"
    assert np.array_equal(r_all.read_hdr(), r_groups.read_hdr()) and np.array_equal(r_all.read_depth(), r_groups.read_depth())
    assert np.array_equal(ro.read_depth(), r_groups.read_depth())
    hp.assert_image_close(out_o, out_g)
    # a second frame on the same context (buffers are reused, counters restart)
    out_g2, _ = r_groups.render_frame(**frame)
    assert np.array_equal(out_g2, out_g)


@pytest.mark.parametrize("size", [(320, 180), (333, 207), (2050, 1030)])
def test_fragment_lists_from_the_rasteriser_or_from_the_visibility_image(hip_capi, monkeypatch, size):
    """The fragment lists of the opaque pass are built inside the 4x rasteriser (from the tile's samples in LDS) or,
    with RDR_FUSED_CLASSIFY=0, by a separate pass over the visibility image. Both must give the same frame bit for
    bit (HDR image, 8-bit frame, depth, visibility), at sizes with partial tiles and with rectangles of unequal tile
    counts, over several frames in a row (the lists exist once per frame slot)."""
    from renderer_amd.api import Renderer
    W, H = size
    frames = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("RDR_FUSED_CLASSIFY", mode)
        r = Renderer(hip_capi, W, H, hp.default_settings(**hp.small_settings(4)))
        scenes.small_scene(r)
        got = []
        for k in range(3):
            r.set_camera(position=(0.3 * k - 0.4, 1.3, 4.0 - 0.3 * k), euler=(-0.12, 0.07 * k, 0.0))
            out, avg = r.render_frame(exposure_mode=1, temporal_luminance=0.5)
            got.append((out.copy(), r.read_hdr(), r.read_depth(), r.read_visibility(), avg))
        frames[mode] = got
        r.close()
    for k in range(3):
        a, b = frames["1"][k], frames["0"][k]
        for x, y, what in zip(a[:4], b[:4], ("frame", "hdr", "depth", "visibility")):
            assert np.array_equal(x, y), f"{what} of frame {k} differs between the two ways of building the lists"
        assert a[4] == b[4]


@pytest.mark.parametrize("big_maps", [False, True])
def test_pair_scan_and_sort_in_one_launch_or_in_several(hip_capi, monkeypatch, big_maps):
    """Below two million elements the pair-count scan is one chained kernel and a digit pass of the pair sort is one
    chained kernel; above, and with RDR_SCAN_CLASSIC=1 / RDR_SORT_CLASSIC=1, they are the multi-kernel versions. The
    tile lists they produce must be the same, so depth, visibility, the shadow maps and the frame are equal bit for
    bit, over several frames on one context (the chained kernels leave their status words for the next launch)."""
    from renderer_amd.api import Renderer
    frames = {}
    for mode in ("0", "1"):
        monkeypatch.setenv("RDR_SCAN_CLASSIC", mode)
        monkeypatch.setenv("RDR_SORT_CLASSIC", mode)
        # big_maps: 2048^2 cascades and eight 1024^2 cubes make more than 2^18 tiles, i.e. three digits of seven bits
        # instead of two of nine
        settings = dict(hp.small_settings(4))
        if big_maps:
            settings.update(directional_shadow_map_resolution=2048, point_shadow_map_resolution=1024, point_shadow_map_count=8)
        r = Renderer(hip_capi, 333, 207, hp.default_settings(**settings))
        scenes.small_scene(r)
        got = []
        for k in range(3):
            r.set_camera(position=(0.3 * k - 0.4, 1.3, 4.0 - 0.3 * k), euler=(-0.12, 0.07 * k, 0.0))
            out, avg = r.render_frame(exposure_mode=1, temporal_luminance=0.5)
            got.append((out.copy(), r.read_hdr(), r.read_depth(), r.read_visibility(), r.read_directional_shadow(0), avg,
                        r.read_point_shadow(1, 3)))
        frames[mode] = got
        r.close()
    for k in range(3):
        a, b = frames["0"][k], frames["1"][k]
        for x, y, what in zip(a[:5] + a[6:], b[:5] + b[6:], ("frame", "hdr", "depth", "visibility", "cascade 0", "cube 1 face -Y")):
            assert np.array_equal(x, y), f"{what} of frame {k} differs between the chained and the multi-kernel scan / sort"
        assert a[5] == b[5]


def test_occlusion_and_metal_roughness_in_one_image(orc_capi, hip_capi, monkeypatch):
    """glTF assets usually pack occlusion (R), roughness (G) and metalness (B) into one image and name it in both
    material slots (main.cpp:135-166 reads the slots independently). The shading kernel then fetches the image once
    per fragment; the frame must equal the checker's, which samples both slots, as closely as with five images."""
    textures = scenes.material_textures

    def packed(r, size, seed, **kw):
        t = list(textures(r, size, seed, **kw))
        t[2] = t[3]  # the grey occlusion image in both slots: occlusion from R, roughness from G, metalness from B, none of them 0
        return t

    monkeypatch.setattr(scenes, "material_textures", packed)
    W, H = 320, 180
    ro, rh = hp.make_pair(orc_capi, hip_capi, W, H, **hp.small_settings(4))
    scenes.small_scene(ro)
    scenes.small_scene(rh)
    out_o, avg_o = ro.render_frame(exposure_mode=1, temporal_luminance=0.5)
    out_h, avg_h = rh.render_frame(exposure_mode=1, temporal_luminance=0.5)
    assert np.array_equal(ro.read_visibility(), rh.read_visibility())
    hp.assert_image_close(out_o, out_h)
    assert avg_o == avg_h or abs(avg_o - avg_h) < 1e-3
    hdr_o, hdr_h = ro.read_hdr(), rh.read_hdr()
    # the colour image to half precision nearly everywhere (a shadow edge may fall on the other side of a sample in a few pixels)
    assert np.mean(np.abs(hdr_o - hdr_h) > 4e-3 * np.abs(hdr_o) + 2e-3) < 1e-3


@pytest.mark.parametrize("size", [(320, 180), (251, 141)])
def test_float_output_frame(orc_capi, hip_capi, size):
    """hdrOutputEnabled (VulkanRenderer.cpp:668-673): the frame arrives as RGBA16F, through the host destination, a
    device destination and rdr_post_process; against the checker within the 8-bit output's bar (1/255), and nearly all
    values within one binary16 step."""
    import torch
    W, H = size
    ro, rh = hp.make_pair(orc_capi, hip_capi, W, H, hdr_output_enabled=1, **hp.small_settings(4))
    scenes.small_scene(ro)
    scenes.small_scene(rh)
    out_o, avg_o = ro.render_frame(exposure_mode=1, temporal_luminance=0.5)
    out_h, avg_h = rh.render_frame(exposure_mode=1, temporal_luminance=0.5)
    assert out_o.dtype == np.float16 and out_h.dtype == np.float16 and out_h.shape == (H, W, 4)
    a, b = out_o.astype(np.float64), out_h.astype(np.float64)
    assert np.all(b[..., 3] == 1.0) and np.all(np.isfinite(b))
    assert np.abs(a - b).max() <= 1.0 / 255.0
    step = np.spacing(np.maximum(out_o, np.float16(2.0 ** -10))).astype(np.float64)
    assert (np.abs(a - b) <= step).mean() > 0.995
    assert abs(avg_o - avg_h) < 1e-3
    # the same frame into device memory: written by the tonemap kernel itself (16-byte aligned destination) or copied
    # (a destination that is not)
    for offset in (0, 8):
        buf = torch.zeros(W * H * 8 + 16, dtype=torch.uint8, device="cuda")
        rh.render_frame(exposure_mode=1, temporal_luminance=0.5, device_ptr=buf.data_ptr() + offset)
        rh.synchronize()
        got = buf[offset:offset + W * H * 8].cpu().numpy().view(np.float16).reshape(H, W, 4)
        assert np.array_equal(got.view(np.uint16), out_h.view(np.uint16)), offset
    # the tail alone on the colour image of that frame
    hdr = rh.read_hdr()
    pp_h, _ = rh.post_process(hdr, rh.frame_stats().exposure)
    assert np.array_equal(pp_h.view(np.uint16), out_h.view(np.uint16))
    ro.close(); rh.close()


@pytest.mark.parametrize("case", [(1, 1, 1, 1, 1), (1, 1, 4, 7, 3), (2, 3, 4, 16, 16), (16, 16, 1, 33, 17), (15, 33, 4, 64, 32)])
def test_smallest_frames_and_shadow_maps(orc_capi, hip_capi, case):
    """The other end of the sizes: a frame of one pixel, frames smaller than one 16-pixel tile, shadow maps of one texel
    and of odd sizes (the hall at a tenth of its detail, all sixteen lights), two frames each."""
    W, H, msaa, dir_res, point_res = case
    ro, rh = hp.make_pair(orc_capi, hip_capi, W, H, msaa=msaa, directional_shadow_map_resolution=dir_res, point_shadow_map_resolution=point_res)
    for r in (ro, rh):
        scenes.sponza_like(r, tex_size=16, env_res=8, detail=0.1)
    for cam in (((-11.0, 2.2, 0.6), (-0.05, -1.45, 0.0)), ((-3.0, 1.0, 0.4), (-0.4, 0.6, 0.0))):
        for r in (ro, rh):
            r.set_camera(position=cam[0], euler=cam[1])
        a, la = ro.render_frame(exposure_mode=1, temporal_luminance=0.5)
        b, lb = rh.render_frame(exposure_mode=1, temporal_luminance=0.5)
        assert np.array_equal(ro.read_depth(), rh.read_depth()) and np.array_equal(ro.read_visibility(), rh.read_visibility())
        for c in range(4):
            assert np.array_equal(ro.read_directional_shadow(c), rh.read_directional_shadow(c)), c
        for slot in range(8):
            for face in range(6):
                assert np.array_equal(ro.read_point_shadow(slot, face), rh.read_point_shadow(slot, face)), (slot, face)
        assert np.abs(a.astype(int) - b.astype(int)).max() <= 1 and abs(la - lb) < 2e-3
    ro.close(); rh.close()


@pytest.mark.parametrize("cubes", [3, 8, 64])
def test_many_point_lights(orc_capi, hip_capi, cubes):
    """The reference's light list has no fixed length (pbr.frag:123 loops over the storage buffer's length): 67 lights, an
    odd count, 12 of them casting shadows, with fewer cube maps than casters (the nearest get them,
    VulkanRendererShadow.cpp:280-298), as many as the default, and more than there are casters."""
    W, H = 320, 200
    ro, rh = hp.make_pair(orc_capi, hip_capi, W, H, msaa=4, directional_shadow_map_resolution=512, point_shadow_map_resolution=128,
                          point_shadow_map_count=cubes)
    rng = np.random.default_rng(5)
    lights = [((float(rng.uniform(-12, 12)), float(rng.uniform(0.4, 5.0)), float(rng.uniform(-5, 5))), tuple(float(v) for v in rng.uniform(2.0, 30.0, 3)), k % 6 == 0)
              for k in range(67)]
    assert sum(l[2] for l in lights) == 12
    for r in (ro, rh):
        scenes.sponza_like(r, tex_size=32, env_res=16, detail=0.2)
        r.set_lights(lights, scenes.default_directional_light())
    for cam in (((-11.0, 2.2, 0.6), (-0.05, -1.45, 0.0)), ((5.0, 1.5, -0.5), (-0.1, 1.2, 0.0))):  # the second camera reassigns the cubes
        for r in (ro, rh):
            r.set_camera(position=cam[0], euler=cam[1])
        a, la = ro.render_frame(exposure_mode=1, temporal_luminance=0.5)
        b, lb = rh.render_frame(exposure_mode=1, temporal_luminance=0.5)
        assert np.array_equal(ro.read_depth(), rh.read_depth()) and np.array_equal(ro.read_visibility(), rh.read_visibility())
        for slot in range(min(cubes, 12)):
            for face in range(6):
                assert np.array_equal(ro.read_point_shadow(slot, face), rh.read_point_shadow(slot, face)), (slot, face)
        hp.assert_image_close(a, b)
        assert abs(la - lb) < 2e-3
    ro.close(); rh.close()
