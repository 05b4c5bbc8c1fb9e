"""Draw lists built on the GPU (renderer_amd/csrc/kernels_draws.hip) against the host loop of the CPU checker, which
follows drawMeshes / shouldCullMesh (Renderer/VulkanRenderer.cpp:929-999) view by view: a scene of 10^4 primitives,
each a mesh of its own, with dynamic nodes, blended and alpha-masked materials and equal barycentre distances."""
import numpy as np
import pytest

from renderer_amd import scenes
from renderer_amd.api import Renderer, default_settings
from tests import helpers as hp

pytestmark = pytest.mark.gpu

SETTINGS = dict(msaa=4, directional_shadow_map_resolution=512, point_shadow_map_resolution=128)
FRAME = dict(exposure_mode=1, temporal_luminance=0.5)


def _move(r, ids, t):
    """Frame t > 0: the camera leaves the symmetry plane and a few dynamic boxes move."""
    r.set_camera(position=(0.8 * t, 6.0 - 0.4 * t, ids["half"] + 6.0 - 2.0 * t), euler=(-0.45, 0.06 * t, 0.0))
    for n, (mid, ix, iz) in enumerate(ids["dynamic"]):
        if n % 5 == 0:
            y = 0.3 + 0.2 * ((n + t) % 4)
            r.set_mesh_transform(mid, scenes.translation(ix * ids["pitch"] - ids["half"], y, iz * ids["pitch"] - ids["half"]).T)


def test_ten_thousand_primitives_match_the_host_loop(orc_capi, hip_capi, monkeypatch):
    W, H = 480, 270
    ro = Renderer(orc_capi, W, H, default_settings(**SETTINGS))
    rd = Renderer(hip_capi, W, H, default_settings(**SETTINGS))  # 10^4 primitives: the draw lists are built on the GPU
    rh = Renderer(hip_capi, W, H, default_settings(**SETTINGS))  # same library, lists built by its host loop
    ids = [scenes.many_primitives(r, side=100) for r in (ro, rd, rh)]
    for t in range(3):
        if t:
            for r, i in zip((ro, rd, rh), ids):
                _move(r, i, t)
        out_o, avg_o = ro.render_frame(**FRAME)
        monkeypatch.delenv("RDR_DEVICE_DRAW_LISTS", raising=False)
        out_d, avg_d = rd.render_frame(**FRAME)
        monkeypatch.setenv("RDR_DEVICE_DRAW_LISTS", "0")
        out_h, avg_h = rh.render_frame(**FRAME)
        monkeypatch.delenv("RDR_DEVICE_DRAW_LISTS", raising=False)
        so, sd, sh = ro.frame_stats(), rd.frame_stats(), rh.frame_stats()
        for s in (sd, sh):
            assert (s.triangle_instances, s.triangles_binned, s.views, s.draws_main) == (so.triangle_instances, so.triangles_binned, so.views, so.draws_main), t
        assert sd.tile_pairs == sh.tile_pairs
        vis_o = ro.read_visibility()
        assert np.array_equal(vis_o, rd.read_visibility()) and np.array_equal(vis_o, rh.read_visibility()), f"visibility ids, frame {t}"
        dep_o = ro.read_depth()
        assert np.array_equal(dep_o, rd.read_depth()) and np.array_equal(dep_o, rh.read_depth())
        for i in range(4):
            c = ro.read_directional_shadow(i)
            assert np.array_equal(c, rd.read_directional_shadow(i)) and np.array_equal(c, rh.read_directional_shadow(i)), f"cascade {i}, frame {t}"
        for slot in range(2):
            for face in range(6):
                p = ro.read_point_shadow(slot, face)
                assert np.array_equal(p, rd.read_point_shadow(slot, face)) and np.array_equal(p, rh.read_point_shadow(slot, face)), f"cube {slot}/{face}, frame {t}"
        assert np.array_equal(out_d, out_h) and avg_d == avg_h, "both ways of building the lists give the same frame"
        hp.assert_image_close(out_o, out_d)
        assert (vis_o != 0xFFFFFFFF).mean() > 0.5
    assert so.draws_main > 2000


def test_ten_thousand_primitives_host_time_at_4k(hip_capi, monkeypatch):
    """At 3840x2160 with 10^4 primitives the host's share of a frame (time inside rdr_render_frame not spent waiting
    for the GPU) must stay below the GPU's frame time when the lists are built on the GPU; with the host loop it does
    not."""
    import torch
    W, H, n = 3840, 2160, 8
    res = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("RDR_DEVICE_DRAW_LISTS", mode)
        r = Renderer(hip_capi, W, H, default_settings(msaa=4))
        ids = scenes.many_primitives(r, side=100)
        dst = [torch.empty((H, W, 4), dtype=torch.uint8, device="cuda:0") for _ in range(2)]
        host, gpu = [], []
        for t in range(n):
            _move(r, ids, t % 3)
            r.render_frame(device_ptr=dst[t & 1].data_ptr(), **FRAME)
            if t >= 2:
                s = r.frame_stats()  # of the frame before: the newest one whose results were picked up
                host.append(s.host_ms); gpu.append(s.ms_total)
        r.synchronize()
        first = dst[(n - 1) & 1].cpu().numpy().copy()
        res[mode] = (float(np.median(host)), float(np.median(gpu)), first)
        r.close()
    print(f"10^4 primitives at 4K: lists on the GPU: host {res['1'][0]:.2f} ms, GPU {res['1'][1]:.2f} ms per frame; host loop: host {res['0'][0]:.2f} ms, GPU {res['0'][1]:.2f} ms")
    assert np.array_equal(res["1"][2], res["0"][2])
    assert res["1"][0] < res["1"][1], "host time below GPU time with the lists built on the GPU"
    assert res["1"][0] < res["0"][0]


def test_unchanged_shadow_views_are_not_rendered_again(orc_capi, hip_capi):
    """Frames of a scene at rest: the cascades of a frame slot and the point-light faces hold what the frame would
    render and are left alone (the reference redraws them, VulkanRendererShadow.cpp:348, 424 - same content). Every
    frame still equals the CPU checker's, which renders everything every time; moving a mesh or the camera brings
    the views back."""
    W, H = 320, 180
    ro, rh = hp.make_pair(orc_capi, hip_capi, W, H, **hp.small_settings(4))
    ido, idh = scenes.small_scene(ro), scenes.small_scene(rh)
    seen = []
    for t in range(7):
        if t == 4:   # a dynamic mesh moves: cascades and point-light faces are redrawn
            for r, i in ((ro, ido), (rh, idh)):
                r.set_mesh_transform(i["sphere"], scenes.translation(0.3, 1.1, 0.2).T)
        if t == 6:   # the camera moves: the cascades follow it, the point-light faces do not depend on it
            for r in (ro, rh):
                r.set_camera(position=(0.6, 1.6, 5.0), euler=(-0.2, 0.1, 0.0))
        out_o, avg_o = ro.render_frame(**FRAME)
        out_h, avg_h = rh.render_frame(**FRAME)
        s = rh.frame_stats()
        seen.append((int(s.views), int(s.views_reused)))
        for i in range(4):
            assert np.array_equal(ro.read_directional_shadow(i), rh.read_directional_shadow(i)), (t, i)
        for slot in range(2):
            for face in range(6):
                assert np.array_equal(ro.read_point_shadow(slot, face), rh.read_point_shadow(slot, face)), (t, slot, face)
        assert np.array_equal(ro.read_visibility(), rh.read_visibility()) and np.array_equal(ro.read_depth(), rh.read_depth())
        hp.assert_image_close(out_o, out_h)
    views, reused = zip(*seen)
    assert reused[0] == 0 and reused[1] == 12            # second frame: the 2 x 6 faces already hold the moving sphere's shadow
    assert reused[2] == 12 and reused[3] == 16           # from the fourth on also the 4 cascades of the frame slot (three slots)
    assert reused[4] == 0 and reused[5] == 12            # the sphere moved; one frame later the faces are current again
    assert reused[6] == 12                               # camera moved: cascades redrawn, faces kept
    assert views[3] == 1 and views[0] > views[1] == views[2] > views[3]


def test_slot_changes_between_frames_in_flight(hip_capi):
    """The camera jumps between two spots every frame, so that the two shadow-casting lights swap cube slots and their
    static faces are rendered again each time, while the frame before - whose faces with the moving sphere start from
    those same static faces - may still be on the GPU. Frames kept in flight (device destinations, no wait in between)
    must equal the same frames rendered one at a time."""
    import torch
    W, H, n = 320, 180, 10
    spots = [dict(position=(2.4, 1.7, 4.0), euler=(-0.2, 0.3, 0.0)), dict(position=(-3.0, 1.6, 4.2), euler=(-0.2, -0.35, 0.0))]
    rp = Renderer(hip_capi, W, H, default_settings(**SETTINGS))
    rs = Renderer(hip_capi, W, H, default_settings(**SETTINGS))
    ip, isr = scenes.small_scene(rp), scenes.small_scene(rs)
    dst = [torch.empty((H, W, 4), dtype=torch.uint8, device="cuda:0") for _ in range(n)]
    serial = []
    for t in range(n):
        for r, ids in ((rp, ip), (rs, isr)):
            r.set_camera(**spots[t & 1])
            r.set_mesh_transform(ids["sphere"], scenes.translation(0.1 * t, 0.9, 0.4).T)
        rp.render_frame(device_ptr=dst[t].data_ptr(), **FRAME)
        out, _ = rs.render_frame(**FRAME)
        serial.append(out.copy())
    rp.synchronize()
    for t in range(n):
        assert np.array_equal(dst[t].cpu().numpy(), serial[t]), f"frame {t}"
    # the slots did change hands: the nearest light differs between the two spots
    d = [[np.linalg.norm(np.subtract(l, s["position"])) for l in ((1.5, 2.2, 1.5), (-2.0, 1.8, 2.0))] for s in spots]
    assert (d[0][0] < d[0][1]) != (d[1][0] < d[1][1])
