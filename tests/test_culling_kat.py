"""Known answers for mesh-level frustum culling (which primitives every view of a frame draws).

Float64 numpy written from the reference's source, not from oracle/ and not from the library:
  VulkanRenderer.cpp:929-944     shouldCullMesh: the six planes of the view in the primitive's local space, the box
                                 corner furthest along each plane's normal, culled if that corner is behind a plane
  Camera.cpp:245-298             the planes from the rows of viewProj * model (pointing into the frustum)
  VulkanRendererShadow.cpp:424-430  the same test per cascade, against the cascade's cropped light camera
The cascade cameras come from tests/test_cascade_kat.py (float64 from the source as well).

A frame's `triangle_instances` is the number of triangles of all draws of all its views: with lights that cast no
point shadows those are the four cascades and the main view, so the count is known from the scene. The boxes are
placed so that every (view, primitive) decision has a margin: in view, behind the camera, beyond the far plane, off
to each side, straddling a side plane (kept), rotated about their own node. The same assertions run on the CPU checker
and, under -m gpu, on the HIP library (host loop and, forced, the draw lists built on the GPU)."""
import numpy as np
import pytest

from renderer_amd import _binding as B
from renderer_amd import scenes
from renderer_amd.api import Renderer, default_settings
from tests.test_cascade_kat import FLIP, expected_cascades, perspective_rh_zo, view_matrix


def planes_of(m):
    """Camera.cpp:245-283: rows of the matrix combined; m in the maths convention (m @ column vector)."""
    return np.array([m[3] + m[0], m[3] - m[0], m[3] + m[1], m[3] - m[1], m[3] - m[2], m[2]])


def culled(view_proj, model, bb_min, bb_max):
    worst = np.inf
    out = False
    for pl in planes_of(view_proj @ model):
        v = np.where(pl[:3] >= 0.0, bb_max, bb_min)
        dist = float(v @ pl[:3] + pl[3])
        worst = min(worst, abs(dist) / max(np.linalg.norm(pl[:3]), 1e-30))
        if dist < 0.0:
            out = True
    return out, worst


def build(r):
    """Returns [(model, bb_min, bb_max, triangles)] of every primitive, in maths convention."""
    tex = scenes.material_textures(r, 8, 5)
    mat = scenes.make_material(tex)
    prims = []

    def add(geo, model):
        p = scenes.upload_primitive(r, geo, mat, B.UINT16)
        r.add_mesh([p], model=model.T)
        prims.append((model.astype(np.float64), np.asarray(p["bb_min"], np.float64), np.asarray(p["bb_max"], np.float64), geo.idx.size // 3))

    add(scenes.plane(40, 40, 1, 1), np.eye(4))
    T, Ry, Rx = scenes.translation, scenes.rotation_y, scenes.rotation_x
    places = [
        T(0.0, 1.0, -6.0),                      # straight ahead
        T(1.5, 0.5, -3.0) @ Ry(0.7),            # ahead, turned about its node
        T(0.0, 1.0, 6.0),                       # behind the camera
        T(0.0, 1.0, -45.0),                     # beyond the far plane of the main view
        T(-30.0, 1.0, -10.0),                   # far off to the left
        T(30.0, 1.0, -10.0) @ Rx(0.4),          # far off to the right
        T(0.0, 25.0, -10.0),                    # far above
        T(-3.9, 1.0, -6.0),                     # across the left side plane: kept
        T(0.0, 0.6, -0.6),                      # across the near plane: kept
        T(6.0, 1.0, -20.0) @ Ry(1.1),           # ahead, to the right, far
        T(-2.0, 3.0, -12.0),
        T(0.5, 0.5, -28.0),                     # just inside the far plane
        T(12.0, 1.0, 3.0),                      # beside and behind
        T(-1.0, 1.0, -9.0) @ Ry(-0.4) @ Rx(0.3),
    ]
    sizes = [(1.0, 2.0, 1.0), (0.8, 0.8, 0.8), (1.5, 1.0, 0.7)]
    for k, m in enumerate(places):
        add(scenes.box(*sizes[k % 3], 1), m)
    return prims


def check_culling(capi, monkeypatch=None, device_lists=None):
    if device_lists is not None:
        monkeypatch.setenv("RDR_DEVICE_DRAW_LISTS", device_lists)
    W, H = 128, 72
    cam_pos, cam_euler, fov, near, far = (0.0, 1.6, 0.0), (-0.05, 0.1, 0.0), np.radians(50.0), 0.1, 30.0
    light_pos, light_dir, up = (0.0, 100.0, 0.0), (0.05, -1.0, 0.05), (1.0, 0.0, 0.0)
    r = Renderer(capi, W, H, default_settings(msaa=1, directional_shadow_map_resolution=64, point_shadow_map_resolution=16))
    prims = build(r)
    r.set_brdf_lut(scenes.brdf_lut(16, 16))
    r.set_environment_map(scenes.sky_faces(4))
    r.set_camera(position=cam_pos, euler=cam_euler, near=near, far=far, fov_or_height=fov)
    dirn = np.asarray(light_dir, np.float32)
    dirn /= np.float32(np.linalg.norm(dirn))
    quat = scenes.quat_look_at(dirn, up)
    r.set_lights([((0.0, 3.0, -5.0), (8.0, 8.0, 8.0), False)], (light_pos, (10.0, 8.0, 6.0), quat))
    r.render_frame()
    st = r.frame_stats()
    aspect = float(np.float32(W) / np.float32(H))
    fov32 = float(np.float32(fov))
    cascades, _ = expected_cascades(cam_pos, cam_euler, fov32, aspect, near, far, light_pos, np.asarray(quat, np.float64), 4)
    main_vp = FLIP @ perspective_rh_zo(fov32, aspect, near, far) @ view_matrix(cam_pos, cam_euler)
    views = [m for m, _, _ in cascades] + [main_vp]
    total, drawn_main, tightest = 0, 0, np.inf
    for vi, vp in enumerate(views):
        for model, lo, hi, tris in prims:
            out, margin = culled(vp, model, lo, hi)
            tightest = min(tightest, margin)
            if not out:
                total += tris
                drawn_main += 1 if vi == len(views) - 1 else 0
    assert tightest > 1e-3, "the scene is meant to leave every decision a margin"
    assert 2 < drawn_main < len(prims), "some primitives of the main view are culled and some are not"
    assert st.views == 5
    assert st.draws_main == drawn_main, (st.draws_main, drawn_main)
    assert st.triangle_instances == total, (st.triangle_instances, total)
    r.close()


def test_frustum_culling_against_float64(orc_capi):
    check_culling(orc_capi)


@pytest.mark.gpu
@pytest.mark.parametrize("device_lists", ["0", "1"])
def test_frustum_culling_against_float64_hip(hip_capi, monkeypatch, device_lists):
    check_culling(hip_capi, monkeypatch, device_lists)
