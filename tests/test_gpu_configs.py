"""GPU parity on reduced versions of the BASELINE.json configurations, and size-independent properties at full
size (where the CPU checker would take minutes per frame)."""
import numpy as np
import pytest

from renderer_amd import scenes
from renderer_amd.api import Renderer, default_settings
from tests import helpers as hp

pytestmark = pytest.mark.gpu
FRAME = dict(exposure_mode=1, temporal_luminance=0.5)


def compare_frame(ro, rh, cascades=True):
    out_o, avg_o = ro.render_frame(**FRAME)
    out_h, avg_h = rh.render_frame(**FRAME)
    assert np.array_equal(ro.read_depth(), rh.read_depth()), "depth"
    assert np.array_equal(ro.read_visibility(), rh.read_visibility()), "coverage"
    if cascades:
        for i in range(4):
            assert np.array_equal(ro.read_directional_shadow(i), rh.read_directional_shadow(i)), f"cascade {i}"
    hp.assert_image_close(out_o, out_h)
    assert abs(avg_o - avg_h) < 2e-3
    return out_o, out_h


def test_config1_helmet_like_reduced(orc_capi, hip_capi):
    """Config 1: one textured ~69k-triangle mesh, directional light + cascades, IBL, bloom + tonemap (here 640x360,
    512^2 textures, 1024^2 cascades)."""
    kw = dict(msaa=4, directional_shadow_map_resolution=1024)
    ro, rh = hp.make_pair(orc_capi, hip_capi, 640, 360, **kw)
    for r in (ro, rh):
        scenes.helmet_like(r, tex_size=512, env_res=64)
    compare_frame(ro, rh)
    for which, levels in ((1, 1), (2, 10)):
        for level in range(levels):
            assert np.array_equal(ro.read_env_cube(which, level), rh.read_env_cube(which, level)), (which, level)


@pytest.mark.parametrize("msaa", [1, 4])
def test_config1_helmet_like_at_its_stated_size(orc_capi, hip_capi, msaa):
    """Config 1 as BASELINE.json states it: 1920x1080, the ~69k-triangle object with five 2048^2 textures, directional
    light with the default 4 x 2048^2 cascades, IBL, bloom + tonemap, at 1 and 4 samples per pixel, from two
    cameras (the second close enough for the textures to be magnified over part of the object)."""
    ro, rh = hp.make_pair(orc_capi, hip_capi, 1920, 1080, msaa=msaa)
    for r in (ro, rh):
        scenes.helmet_like(r, tex_size=2048)
    compare_frame(ro, rh)
    vis = rh.read_visibility()
    assert 0.15 < (vis[0] != 0xFFFFFFFF).mean() < 0.9, "the object covers part of the frame"
    for r in (ro, rh):
        r.set_camera(position=(0.9, 0.5, 1.5), euler=(-0.25, 0.55, 0.0))
    compare_frame(ro, rh)


def test_config2_sponza_like_reduced(orc_capi, hip_capi):
    """Config 2: the hall with 16 point lights (8 shadowed), MASK and BLEND prims, at quarter detail and 640x360."""
    kw = dict(msaa=4, directional_shadow_map_resolution=1024, point_shadow_map_resolution=256)
    ro, rh = hp.make_pair(orc_capi, hip_capi, 640, 360, **kw)
    for r in (ro, rh):
        scenes.sponza_like(r, tex_size=128, env_res=32, detail=0.25)
    compare_frame(ro, rh)
    for slot in (0, 3, 7):
        for face in (0, 2, 5):
            assert np.array_equal(ro.read_point_shadow(slot, face), rh.read_point_shadow(slot, face)), (slot, face)
    # second frame from another camera: static point-shadow maps are reused, dynamic props re-rendered
    for r in (ro, rh):
        r.set_camera(position=(-6.0, 2.5, 0.2), euler=(-0.05, -1.3, 0.0))
    compare_frame(ro, rh)


def test_config3_animation_batch_reduced(orc_capi, hip_capi):
    """Config 3: an animation batch with a skinned figure (key-framed joints, t_k = k / 30 s scaled up so that six
    frames span the clip) and a node-TRS track (the reference's only animation form). Every frame must match the
    checker, and frame order must not matter with a fixed adapted luminance."""
    ro, rh = hp.make_pair(orc_capi, hip_capi, 320, 180, **hp.small_settings(4))
    ids_o, ids_h = scenes.small_scene(ro), scenes.small_scene(rh)
    fig_o, fig_h = (scenes.SkinnedFigure(r, origin=(0.8, 0.0, 1.5)) for r in (ro, rh))

    def set_frame(k):
        m = scenes.translation(0.3 * np.sin(k * 0.7), 0.9 + 0.1 * k, 0.4 - 0.2 * k) @ scenes.rotation_y(0.3 * k)
        ro.set_mesh_transform(ids_o["sphere"], m.T)
        rh.set_mesh_transform(ids_h["sphere"], m.T)
        fig_o.pose(k * 11 / 30); fig_h.pose(k * 11 / 30)

    frames = []
    for k in range(6):
        set_frame(k)
        _, out_h = compare_frame(ro, rh)
        frames.append(out_h.copy())
    # replay frame 2 after frame 5: identical bytes
    set_frame(2)
    again, _ = rh.render_frame(**FRAME)
    assert np.array_equal(again, frames[2])


def test_config5_raster_stress_reduced(orc_capi, hip_capi):
    """Config 5: displaced grid of sub-pixel to few-pixel triangles (2 x 400^2 = 320k) from two orbit views:
    binning, the tile walk and the depth test against the checker, bit for bit."""
    ro, rh = hp.make_pair(orc_capi, hip_capi, 512, 288, msaa=4, directional_shadow_map_resolution=512)
    for r in (ro, rh):
        scenes.raster_stress(r, quads_per_side=400)
    for k in (1, 4):
        for r in (ro, rh):
            scenes.orbit_camera(r, k)
        compare_frame(ro, rh)
        so, sh = ro.frame_stats(), rh.frame_stats()
        assert (so.triangle_instances, so.triangles_binned) == (sh.triangle_instances, sh.triangles_binned)


def test_full_size_properties_4k(hip_capi):
    """Sponza-class scene at 3840x2160 msaa4 (the benchmark workload, half detail to keep set-up short):
    rendering is deterministic and order-independent, coverage is consistent with depth, the average luminance is
    the mean of the log-luminance crop, and bloom mip 0 is at least the stored colour."""
    W, H = 3840, 2160
    r = Renderer(hip_capi, W, H, default_settings(msaa=4))
    scenes.sponza_like(r, tex_size=256, env_res=64, detail=0.5)
    lean, avg_lean = r.render_frame(**FRAME)
    lean = lean.copy()
    with pytest.raises(Exception, match="keep_intermediates"):
        r.read_log_luminance()
    with pytest.raises(Exception, match="keep_intermediates"):
        r.read_bloom(0)
    r.keep_intermediates(True)
    a, avg_a = r.render_frame(**FRAME)
    a = a.copy()
    assert np.array_equal(a, lean) and avg_a == avg_lean, "storing the intermediate images must not change the frame"
    depth, vis = r.read_depth(), r.read_visibility()
    assert np.array_equal((vis == 0xFFFFFFFF), (depth >= 1.0)), "a sample is covered exactly when its depth was written"
    assert depth.min() >= 0.0 and depth.max() <= 1.0
    ll = r.read_log_luminance()
    crop = ll[H // 2 - 128:H // 2 + 128, W // 2 - 128:W // 2 + 128].astype(np.float64)
    assert abs(crop.mean() - avg_a) < 1e-4
    hdr = r.read_hdr()
    b0 = r.read_bloom(0)
    assert np.all(b0 >= hdr[..., :3] * (1 - 2.0 ** -5) - 1e-6), "bloom mip 0 = truncated colour + non-negative up-sampled light"
    assert np.all(a[..., 3] == 255)
    r.set_camera(position=(-6.0, 2.5, 0.2), euler=(-0.05, -1.3, 0.0))
    b, _ = r.render_frame(**FRAME)
    assert not np.array_equal(a, b)
    r.set_camera(position=(-11.0, 2.2, 0.6), euler=(-0.05, -1.45, 0.0))
    c, avg_c = r.render_frame(**FRAME)
    assert np.array_equal(a, c) and avg_a == avg_c, "same inputs, same bytes, whatever was rendered in between"
    # exposure mode 2 with the exposure mode 1 computed reproduces the frame
    e = r.frame_stats().exposure
    d, _ = r.render_frame(exposure_mode=2, exposure=e)
    assert np.array_equal(c, d)
    r.close()


def test_config5_full_size_ten_million_triangles(hip_capi):
    """Config 5 at its full size: a 9 999 392-triangle displaced grid (one primitive) from 8 orbit views at
    3840x2160 with 4x MSAA and 4 cascades. The CPU checker would need minutes per view, so the views are held to the
    size-independent properties: a sample is covered exactly when its depth was written, depth stays in [0, 1], the
    counters add up (every view submits the grid once per cascade that sees it plus once for the main view), each view
    differs from the others, and rendering a view again gives the same bytes, depth and counters."""
    W, H = 3840, 2160
    r = Renderer(hip_capi, W, H, default_settings(msaa=4))
    info = scenes.raster_stress(r, quads_per_side=2236)
    assert info["triangles"] == 2 * 2236 * 2236
    frames, stats = [], []
    for k in range(8):
        scenes.orbit_camera(r, k)
        out, _ = r.render_frame(**FRAME)
        s = r.frame_stats()
        stats.append((int(s.triangle_instances), int(s.triangles_binned), int(s.tile_pairs), int(s.views)))
        assert s.triangle_instances % info["triangles"] == 0 and 2 <= s.triangle_instances // info["triangles"] <= 5
        assert 0 < s.triangles_binned < s.triangle_instances and s.tile_pairs >= s.triangles_binned
        depth, vis = r.read_depth(), r.read_visibility()
        assert np.array_equal(vis == 0xFFFFFFFF, depth >= 1.0), f"view {k}: coverage and depth disagree"
        assert depth.min() >= 0.0 and depth.max() <= 1.0
        assert vis[vis != 0xFFFFFFFF].max() < info["triangles"]
        assert (vis[0] != 0xFFFFFFFF).mean() > 0.3, f"view {k}: the grid fills a good part of the frame"
        assert out[..., :3].std() > 5
        frames.append(out.copy())
        if k in (0, 5):
            keep_depth, keep_vis = depth, vis
            again, _ = r.render_frame(**FRAME)
            s2 = r.frame_stats()
            assert np.array_equal(again, frames[k]), f"view {k} is not reproducible"
            assert np.array_equal(r.read_depth(), keep_depth) and np.array_equal(r.read_visibility(), keep_vis)
            assert (int(s2.triangle_instances), int(s2.triangles_binned), int(s2.tile_pairs), int(s2.views)) == stats[k]
    assert len({f.tobytes() for f in frames}) == 8
    r.close()


def test_config3_animation_batch_1080p(hip_capi):
    """Config 3 at 1920x1080 with 4x MSAA on one GPU: 32 frames (t_k = k / 30 s) of the hall at half detail with two
    skinned, key-framed figures, rendered through the frame sharder with two frames in flight. The whole batch is
    rendered twice: every frame must come out the same both times (replay determinism), differ from its neighbours,
    and three frames rendered on their own, to completion, by a second context must equal the batch's."""
    import torch

    from renderer_amd.sharding import FrameSharder

    W, H, n = 1920, 1080, 32

    def build():
        r = Renderer(hip_capi, W, H, default_settings(msaa=4))
        scenes.sponza_like(r, tex_size=256, env_res=64, detail=0.5)
        figs = [scenes.SkinnedFigure(r, n_bones=16, length=3.2, radius=0.3, nu=64, nv=256, tex_size=128, origin=o, seed=61 + i)
                for i, o in enumerate(((-6.0, 0.0, 0.5), (-3.0, 0.0, -1.0)))]
        return r, figs

    def set_frame_for(figs):
        def set_frame(rr, k):
            rr.set_camera(position=(-11.0 + 0.05 * k, 2.2, 0.6), euler=(-0.05, -1.45 + 0.004 * k, 0.0))
            for f in figs:
                f.pose(k / 30.0)
        return set_frame

    r, figs = build()
    sharder = FrameSharder(r, dist=None, device="cuda:0", depth=2)
    params = dict(exposure_mode=1, temporal_luminance=0.5)
    runs = []
    for _ in range(2):
        got = {}
        lum = sharder.run(n, set_frame_for(figs), params, on_frame=lambda k, t: got.__setitem__(k, t.cpu().numpy().copy()))
        torch.cuda.synchronize()
        assert sorted(got) == list(range(n)) and sorted(lum) == list(range(n))
        runs.append((got, lum))
    for k in range(n):
        assert np.array_equal(runs[0][0][k], runs[1][0][k]), f"frame {k} differs between two runs of the batch"
        assert runs[0][1][k] == runs[1][1][k]
    assert len({runs[0][0][k].tobytes() for k in range(n)}) == n
    r.close()
    r2, figs2 = build()
    for k in (0, 13, 31):
        set_frame_for(figs2)(r2, k)
        img, avg = r2.render_frame(**params)
        assert np.array_equal(img, runs[0][0][k]), f"frame {k} rendered alone differs from the batch"
        assert avg == runs[0][1][k]
    r2.close()


def test_benchmark_frame_against_the_checker_at_full_size(orc_capi, hip_capi):
    """The exact frame bench.py times (Sponza-class scene at full detail, 1024^2 textures, 3840x2160, 4x MSAA, 4
    cascades of 2048^2, 16 point lights of which 8 cast shadows, IBL, bloom + tonemap), HIP library against the CPU
    checker: depth, coverage, cascades and point-light faces bit for bit, the 8-bit frame within 1/255 and 45 dB."""
    W, H = 3840, 2160
    ro, rh = hp.make_pair(orc_capi, hip_capi, W, H, msaa=4)
    scenes.sponza_like(ro)
    scenes.sponza_like(rh)
    for r in (ro, rh):  # frame 0 of the benchmark's camera path
        r.set_camera(position=(-11.0, 2.2, 0.6), euler=(-0.05, -1.45, 0.0))
    compare_frame(ro, rh)
    for slot in range(8):
        for face in (0, 3, 5):
            assert np.array_equal(ro.read_point_shadow(slot, face), rh.read_point_shadow(slot, face)), (slot, face)
    so, sh = ro.frame_stats(), rh.frame_stats()
    assert (so.triangle_instances, so.triangles_binned, so.views) == (sh.triangle_instances, sh.triangles_binned, sh.views)
    ro.close(); rh.close()


def test_frames_and_shadow_maps_larger_than_the_benchmark(orc_capi, hip_capi):
    """Beyond the benchmark's size: 7680x4320 at 4 samples per pixel with 4096^2 cascades and 1024^2 cube faces (the hall at
    half detail, its sixteen lights), and a small frame with the largest cascades the settings accept (16384^2; 22 million
    (tile, triangle) pairs, the multi-kernel scan and sort). tools/max_size_check.py goes on to 16384x16384 at 4 samples
    (2^30 of them) and 8192^2 cube faces, which take the checker half a minute a frame."""
    for (W, H, msaa, dir_res, point_res) in ((7680, 4320, 4, 4096, 1024), (512, 512, 1, 16384, 256)):
        ro, rh = hp.make_pair(orc_capi, hip_capi, W, H, msaa=msaa, directional_shadow_map_resolution=dir_res, point_shadow_map_resolution=point_res)
        for r in (ro, rh):
            scenes.sponza_like(r, tex_size=256, detail=0.5)
            r.set_camera(position=(-6.5, 2.2, 0.2), euler=(-0.2, -1.0, 0.0))
        compare_frame(ro, rh)
        for slot in (0, 5):
            for face in (1, 3):
                assert np.array_equal(ro.read_point_shadow(slot, face), rh.read_point_shadow(slot, face)), (W, slot, face)
        ro.close(); rh.close()
