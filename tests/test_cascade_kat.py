"""Known answers for the cascade matrices of the directional light (the host arithmetic of the cascaded shadow pass).

Float64 numpy written from the reference's source, not from oracle/ and not from the library:
  VulkanRendererShadow.cpp:152-175   practical split scheme, lambda 0.9
  VulkanRendererShadow.cpp:402-421   per cascade: the view camera with near / far moved to the split, its eight frustum
                                     corners, their box in the light's clip space, scale 1 / largest side and centre
                                     offset as the light camera's crop matrix
  Camera.cpp:136-138, 196-216        view = Rx(-ex) Ry(-ey) Rz(-ez) T(-p); proj = scale(1,-1,1) * glm::perspective /
                                     glm::ortho(-h a / 2, h a / 2, -h, h, n, f); viewProj = crop * proj * view
  Camera.cpp:218-242, 306-329        getFrustumVertices (inverse of viewProj on the NDC cube with z in [0, 1]),
                                     makeAABBFromVertices
  DirectionalLight.cpp:3-5           the light's camera: orthographic, euler angles of its quaternion, near .1, far 200,
                                     height 10, aspect 1
  Renderer.vcxproj:92                GLM_FORCE_DEPTH_ZERO_TO_ONE: glm::perspectiveRH_ZO, glm::orthoRH_ZO
glm's own formulas (rotate, perspectiveRH_ZO, orthoRH_ZO, eulerAngles of a quaternion) are the published ones.
The same assertions run on the CPU checker and, under -m gpu, on the HIP library (both compute these matrices on the
host, each in its own code)."""
import numpy as np
import pytest

from renderer_amd import scenes
from renderer_amd.api import Renderer, default_settings


def rot(angle, axis):
    c, s = np.cos(angle), np.sin(angle)
    m = np.eye(4)
    i, j = {0: (1, 2), 1: (2, 0), 2: (0, 1)}[axis]
    m[i, i] = c; m[j, j] = c; m[i, j] = -s; m[j, i] = s
    return m


def translate(v):
    m = np.eye(4)
    m[:3, 3] = v
    return m


def scale(v):
    return np.diag([v[0], v[1], v[2], 1.0])


def view_matrix(position, euler):
    return rot(-euler[0], 0) @ rot(-euler[1], 1) @ rot(-euler[2], 2) @ translate(-np.asarray(position, np.float64))


def perspective_rh_zo(fovy, aspect, n, f):
    t = np.tan(fovy / 2.0)
    m = np.zeros((4, 4))
    m[0, 0] = 1.0 / (aspect * t)
    m[1, 1] = 1.0 / t
    m[2, 2] = f / (n - f)
    m[3, 2] = -1.0
    m[2, 3] = -(f * n) / (f - n)
    return m


def ortho_rh_zo(l, r, b, t, n, f):
    m = np.eye(4)
    m[0, 0] = 2.0 / (r - l)
    m[1, 1] = 2.0 / (t - b)
    m[2, 2] = -1.0 / (f - n)
    m[0, 3] = -(r + l) / (r - l)
    m[1, 3] = -(t + b) / (t - b)
    m[2, 3] = -n / (f - n)
    return m


FLIP = scale((1.0, -1.0, 1.0))


def euler_of_quat(q):
    """glm::eulerAngles(quat): (pitch, yaw, roll), q = (w, x, y, z)."""
    w, x, y, z = [float(v) for v in q]
    pitch = np.arctan2(2.0 * (y * z + w * x), w * w - x * x - y * y + z * z)
    yaw = np.arcsin(np.clip(-2.0 * (x * z - w * y), -1.0, 1.0))
    roll = np.arctan2(2.0 * (x * y + w * z), w * w + x * x - y * y - z * z)
    return np.array([pitch, yaw, roll])


def expected_cascades(cam_pos, cam_euler, fov, aspect, near, far, light_pos, light_quat, levels):
    # split distances
    d = [near]
    for i in range(1, levels):
        c_uni = near + (far - near) * i / levels
        c_log = near * (far / near) ** (i / levels)
        d.append(0.9 * c_log + 0.1 * c_uni)
    depths = [0.0] + [(far - far * near / x) / (far - near) for x in d[1:]]
    # the light's camera
    light_view = view_matrix(light_pos, euler_of_quat(light_quat))
    light_proj = FLIP @ ortho_rh_zo(-10.0 * 1.0 / 2, 10.0 * 1.0 / 2, -10.0, 10.0, 0.1, 200.0)
    light_vp = light_proj @ light_view
    cam_view = view_matrix(cam_pos, cam_euler)
    ndc = np.array([[x, y, z, 1.0] for z in (0.0, 1.0) for y in (-1.0, 1.0) for x in (-1.0, 1.0)])
    out = []
    for i in range(levels):
        n_i = d[i]
        f_i = d[i + 1] if i < levels - 1 else far
        split_vp = FLIP @ perspective_rh_zo(fov, aspect, n_i, f_i) @ cam_view
        corners = (np.linalg.inv(split_vp) @ ndc.T).T
        corners = corners[:, :3] / corners[:, 3:4]
        in_light = (light_vp @ np.c_[corners, np.ones(8)].T).T
        in_light = in_light[:, :3] / in_light[:, 3:4]
        lo, hi = in_light.min(0), in_light.max(0)
        largest = max(hi[0] - lo[0], hi[1] - lo[1])
        offset = -np.array([(hi[0] + lo[0]) / 2, (hi[1] + lo[1]) / 2, 0.0])
        crop = scale((1.0 / largest, 1.0 / largest, 1.0)) @ translate(offset)
        out.append((crop @ light_vp, corners, largest))
    return out, depths


CASES = [
    # camera position, euler, vertical field of view, near, far, light position, light direction, up
    ((0.4, 1.5, 3.5), (-0.15, 0.3, 0.0), np.radians(40.0), 0.1, 60.0, (0.0, 100.0, 0.0), (0.05, -1.0, 0.05), (1.0, 0.0, 0.0)),
    ((-6.0, 2.5, 0.5), (-0.05, -1.45, 0.0), np.radians(55.0), 0.25, 35.0, (3.0, 40.0, -2.0), (0.3, -0.8, -0.4), (0.0, 0.0, 1.0)),
    ((2.0, 0.8, -4.0), (0.2, 2.6, 0.1), np.radians(70.0), 0.1, 120.0, (0.0, 80.0, 0.0), (-0.6, -0.7, 0.2), (0.0, 1.0, 0.0)),
]


def check_cascades(capi):
    for cam_pos, cam_euler, fov, near, far, light_pos, light_dir, up in CASES:
        W, H = 96, 54
        r = Renderer(capi, W, H, default_settings(msaa=1, directional_shadow_map_resolution=64, point_shadow_map_resolution=16))
        scenes.small_scene(r, tex_size=8, env_res=4)
        r.set_camera(position=cam_pos, euler=cam_euler, near=near, far=far, fov_or_height=fov)
        dirn = np.asarray(light_dir, np.float32)
        dirn /= np.float32(np.linalg.norm(dirn))
        quat = scenes.quat_look_at(dirn, up)
        r.set_lights([((0.0, 2.0, 0.0), (5.0, 5.0, 5.0), False)], (light_pos, (10.0, 8.0, 6.0), quat))
        r.render_frame()
        vps, depth = r.read_cascades()
        aspect = float(np.float32(W) / np.float32(H))
        want, want_depth = expected_cascades(cam_pos, cam_euler, float(np.float32(fov)), aspect, near, far, light_pos,
                                             np.asarray(quat, np.float64), 4)
        assert np.allclose(depth, want_depth, rtol=3e-6, atol=1e-7)
        for i, (m, corners, largest) in enumerate(want):
            got = vps[i].T.astype(np.float64)  # stored column-major
            # single-precision host arithmetic through a 4x4 inverse: compare what the matrix does to the split's own
            # corners (light clip coordinates in [-0.5, 0.5] by construction) rather than its entries one by one
            p_want = (m @ np.c_[corners, np.ones(8)].T).T
            p_got = (got @ np.c_[corners, np.ones(8)].T).T
            assert np.allclose(p_got[:, 3], 1.0, atol=1e-6)
            assert np.abs(p_got[:, :2] - p_want[:, :2]).max() < 2e-4, (i, np.abs(p_got - p_want).max())
            assert np.abs(p_got[:, 2] - p_want[:, 2]).max() < 2e-5
            # the fitted box: the split's corners span exactly one unit along its larger side, centred on the origin
            span = p_got[:, :2].max(0) - p_got[:, :2].min(0)
            assert abs(span.max() - 1.0) < 5e-4
            assert np.abs(p_got[:, :2].max(0) + p_got[:, :2].min(0)).max() < 5e-4
            # and the scale itself (1 / largest side of the box in the light's clip space, no factor 2: the reference's)
            assert abs(np.linalg.norm(got[0, :3]) / np.linalg.norm(m[0, :3]) - 1.0) < 2e-4
        r.close()


def test_cascade_matrices_against_float64(orc_capi):
    check_cascades(orc_capi)


@pytest.mark.gpu
def test_cascade_matrices_against_float64_hip(hip_capi):
    check_cascades(hip_capi)
