"""Skinning and key-frame interpolation.

The reference has neither skins nor correct interpolation (Renderer/Animation.h:86-98 divides by the wrong quantity
for LINEAR and returns the nearest key for CUBICSPLINE; Renderer/main.cpp:61-177 never reads JOINTS_0 / WEIGHTS_0), so
nothing here can be pinned by it: the interpolation is checked against the closed forms of the glTF 2.0
specification, the CPU skinning against float64 numpy, and the HIP kernel against the CPU skinning bit for bit."""
import numpy as np
import pytest

from renderer_amd import _binding as B
from renderer_amd import animation as A
from renderer_amd import scenes
from renderer_amd.api import Renderer, RendererError, default_settings
from tests import helpers as hp

FRAME = dict(exposure_mode=1, temporal_luminance=0.5)
ERR_INVALID = 1


# ------------------------------------------------------------------ key frames

def test_linear_uses_the_fraction_of_the_key_interval():
    a = A.Animation([1.0, 3.0, 4.0], [[0.0, 0.0, 0.0], [2.0, 4.0, 8.0], [2.0, 4.0, 0.0]])
    assert np.allclose(a.value_at(2.0), [1.0, 2.0, 4.0])      # (t - t0) / (t1 - t0) = 1/2, not t / (t1 - t0) = 1
    assert np.allclose(a.value_at(3.5), [2.0, 4.0, 4.0])
    assert np.allclose(a.value_at(0.0), [0.0, 0.0, 0.0]) and np.allclose(a.value_at(9.0), [2.0, 4.0, 0.0])  # clamp


def test_rotation_keys_are_slerped_along_the_short_arc():
    q0, q1 = A.quat_from_axis_angle((0, 0, 1), 0.0), A.quat_from_axis_angle((0, 0, 1), np.pi / 2)
    a = A.Animation([0.0, 1.0], [q0, q1], rotation=True)
    assert np.allclose(a.value_at(0.5), A.quat_from_axis_angle((0, 0, 1), np.pi / 4))
    assert np.allclose(a.value_at(0.25), A.quat_from_axis_angle((0, 0, 1), np.pi / 8))
    b = A.Animation([0.0, 1.0], [q0, -q1], rotation=True)     # same rotation, opposite sign: still the short way
    assert np.allclose(A.quat_to_matrix(b.value_at(0.5)), A.quat_to_matrix(a.value_at(0.5)))
    assert abs(np.linalg.norm(a.value_at(0.3)) - 1.0) < 1e-12


def test_step_holds_the_previous_key():
    a = A.Animation([0.0, 1.0, 2.0], [[1.0], [5.0], [9.0]], A.STEP)
    assert a.value_at(0.99)[0] == 1.0 and a.value_at(1.0)[0] == 5.0 and a.value_at(1.7)[0] == 5.0 and a.value_at(2.0)[0] == 9.0


def test_cubic_spline_reproduces_a_cubic():
    """With the tangents of p(t) = t^3 - 2t the Hermite segment is p itself."""
    p, dp = (lambda t: t ** 3 - 2 * t), (lambda t: 3 * t * t - 2)
    times = [0.0, 1.5, 2.0]
    rows = []
    for t in times:
        rows += [[dp(t)], [p(t)], [dp(t)]]
    a = A.Animation(times, rows, A.CUBICSPLINE)
    for t in (0.0, 0.4, 1.1, 1.5, 1.8, 2.0):
        assert abs(a.value_at(t)[0] - p(t)) < 1e-12
    with pytest.raises(ValueError):
        A.Animation(times, [[0.0]] * 3, A.CUBICSPLINE)


def test_repeat_modes():
    a = A.Animation([1.0, 3.0], [[0.0], [2.0]], repeat=A.REPEAT)
    assert np.isclose(a.value_at(3.5)[0], 0.5) and np.isclose(a.value_at(0.5)[0], 1.5)
    m = A.Animation([1.0, 3.0], [[0.0], [2.0]], repeat=A.MIRROR)
    assert np.isclose(m.value_at(3.5)[0], 1.5) and np.isclose(m.value_at(5.5)[0], 0.5)
    assert A.Animation([2.0], [[7.0]]).value_at(100.0)[0] == 7.0


def test_node_hierarchy_and_skin_palette():
    child = A.Node(translation=(0, 1, 0))
    root = A.Node(translation=(1, 0, 0), rotation=A.quat_from_axis_angle((0, 0, 1), np.pi / 2), children=[child])
    root.recalculate_model()
    assert np.allclose(child.model @ [0, 0, 0, 1], [0, 0, 0, 1])     # (0,1,0) turned 90 degrees about z is (-1,0,0); +1 in x
    skin = A.Skin([root, child], [np.linalg.inv(root.model), np.linalg.inv(child.model)])
    assert np.allclose(skin.joint_matrices(), np.eye(4)[None])          # bind pose: identity palette
    root.set_transform((2, 0, 0), root.rotation, (1, 1, 1))
    root.recalculate_model()
    assert np.allclose(skin.joint_matrices()[1][:3, 3], [1, 0, 0])       # the child follows its parent
    assert child.is_static()
    child.rotation_animation = A.Animation([0.0], [[0, 0, 0, 1]], rotation=True)
    assert not child.is_static()


# ------------------------------------------------------------------ CPU skinning

def figure_scene(capi, W=160, H=120, msaa=1):
    r = Renderer(capi, W, H, default_settings(msaa=msaa, directional_shadow_map_resolution=256, point_shadow_map_resolution=64))
    scenes.small_scene(r, with_blend=True)
    fig = scenes.SkinnedFigure(r, origin=(0.8, 0.0, 1.5))
    return r, fig


def numpy_skin(geo_pos, joints, weights, palette_abi):
    J = palette_abi.astype(np.float64).transpose(0, 2, 1)                 # back to math convention
    S = np.einsum("vk,vkij->vij", weights.astype(np.float64), J[joints])
    p = np.concatenate([geo_pos.astype(np.float64), np.ones((geo_pos.shape[0], 1))], 1)
    return np.einsum("vij,vj->vi", S, p)[:, :3]


def test_cpu_skinning_against_float64(orc_capi):
    r, fig = figure_scene(orc_capi)
    rest = r.read_primitive_positions(fig.mesh, 0).copy()
    joints, weights = scenes.bone_chain_weights(rest[:, 1].astype(np.float64), 6, 2.4)
    for t in (0.0, 0.37, 1.2, 2.9):
        pal = fig.palette(t)
        r.set_mesh_joints(fig.mesh, pal)
        got = r.read_primitive_positions(fig.mesh, 0)
        want = numpy_skin(rest, joints, weights, pal)
        assert np.abs(got - want).max() < 2e-6, t
    assert np.abs(r.read_primitive_positions(fig.mesh, 0) - rest).max() > 0.1, "the pose moved"
    r.close()


def test_one_rigid_matrix_for_every_joint_is_a_rigid_motion(orc_capi):
    r, fig = figure_scene(orc_capi)
    rest = r.read_primitive_positions(fig.mesh, 1).copy()
    M = scenes.translation(0.3, -0.2, 0.1) @ scenes.rotation_y(0.7) @ scenes.rotation_x(-0.4)
    r.set_mesh_joints(fig.mesh, np.tile(M.T.astype(np.float32), (6, 1, 1)))
    got = r.read_primitive_positions(fig.mesh, 1)
    want = rest.astype(np.float64) @ M[:3, :3].T + M[:3, 3]
    assert np.abs(got - want).max() < 0.02, "8-bit weights sum to 1 within 1/255"
    limb = r.read_primitive_positions(fig.mesh, 0)   # float weights: (1 - u) + u is 1 to an ulp
    assert np.abs(limb - (scenes.cylinder(0.22, 2.4, 24, 48).transformed(scenes.translation(0, 1.2, 0)).pos.astype(np.float64)
                           @ M[:3, :3].T + M[:3, 3])).max() < 1e-5
    r.close()


def test_skinning_argument_checks(orc_capi):
    r, fig = figure_scene(orc_capi)
    pal = fig.palette(0.0)
    assert orc_capi.set_mesh_joints(r._ctx, fig.mesh, B.fptr(pal), 5) == ERR_INVALID        # joint 5 is in use
    assert orc_capi.set_mesh_joints(r._ctx, 0, B.fptr(pal), 6) == ERR_INVALID               # the floor has no skin
    assert orc_capi.set_mesh_joints(r._ctx, 999, B.fptr(pal), 6) == ERR_INVALID
    assert orc_capi.read_primitive_positions(r._ctx, fig.mesh, 7, None, None) == ERR_INVALID
    geo = scenes.cylinder(0.1, 1.0, 8, 4)
    j, w = scenes.bone_chain_weights(geo.pos[:, 1].astype(np.float64) + 0.5, 2, 1.0)
    prim = scenes.upload_skinned_primitive(r, geo, j, w, scenes.make_material())
    with pytest.raises(RendererError):
        r.add_mesh([prim], is_static=True)                                                   # a skinned mesh cannot be static
    r.close()


def test_posed_frame_differs_and_rest_pose_is_the_default(orc_capi):
    r, fig = figure_scene(orc_capi)
    a, _ = r.render_frame(**FRAME)
    a = a.copy()
    fig.pose(0.0)
    b, _ = r.render_frame(**FRAME)
    assert np.abs(a.astype(int) - b.astype(int)).max() <= 1, "t = 0 is the bind pose"
    fig.pose(0.6)
    c, _ = r.render_frame(**FRAME)
    assert (np.abs(a.astype(int) - c.astype(int)).max(-1) > 8).mean() > 0.01
    r.close()


# ------------------------------------------------------------------ GPU parity

@pytest.mark.gpu
def test_hip_skinned_positions_match_bit_for_bit(orc_capi, hip_capi):
    ro, fo = figure_scene(orc_capi)
    rh, fh = figure_scene(hip_capi)
    for t in (0.0, 0.2, 0.77, 1.31, 2.5):
        pal = fo.palette(t)
        ro.set_mesh_joints(fo.mesh, pal); rh.set_mesh_joints(fh.mesh, pal)
        for prim in (0, 1):
            assert np.array_equal(ro.read_primitive_positions(fo.mesh, prim), rh.read_primitive_positions(fh.mesh, prim)), (t, prim)
    ro.close(); rh.close()


@pytest.mark.gpu
def test_hip_skinned_frames_match(orc_capi, hip_capi):
    ro, fo = figure_scene(orc_capi, 320, 180, 4)
    rh, fh = figure_scene(hip_capi, 320, 180, 4)
    frames = {}
    for t in (0.0, 0.45, 1.1, 1.75):
        fo.pose(t); fh.pose(t)
        out_o, avg_o = ro.render_frame(**FRAME)
        out_h, avg_h = rh.render_frame(**FRAME)
        assert np.array_equal(ro.read_depth(), rh.read_depth()), ("depth", t)
        assert np.array_equal(ro.read_visibility(), rh.read_visibility()), ("coverage", t)
        for i in range(4):
            assert np.array_equal(ro.read_directional_shadow(i), rh.read_directional_shadow(i)), ("cascade", i, t)
        for slot in (0, 1):
            assert np.array_equal(ro.read_point_shadow(slot, 3), rh.read_point_shadow(slot, 3)), ("point shadow", slot, t)
        hp.assert_image_close(out_o, out_h)
        frames[t] = out_h.copy()
    fh.pose(0.45)
    again, _ = rh.render_frame(**FRAME)
    assert np.array_equal(again, frames[0.45]), "a pose is a pure function of its palette"
    ro.close(); rh.close()


@pytest.mark.gpu
def test_hip_large_skinned_mesh(hip_capi):
    """A 1M-vertex limb (the size at which the kernel is bandwidth-bound) against float64 numpy."""
    r = Renderer(hip_capi, 64, 64, default_settings(msaa=1))
    geo = scenes.cylinder(0.3, 4.0, 1023, 1023).transformed(scenes.translation(0, 2.0, 0))
    joints, weights = scenes.bone_chain_weights(geo.pos[:, 1].astype(np.float64), 32, 4.0)
    mesh = r.add_mesh([scenes.upload_skinned_primitive(r, geo, joints, weights, scenes.make_material())], is_static=False)
    rng = np.random.default_rng(5)
    pal = np.stack([(scenes.translation(*rng.normal(0, 0.1, 3)) @ scenes.rotation_y(rng.normal(0, 0.3)) @ scenes.rotation_x(rng.normal(0, 0.3))).T
                    for _ in range(32)]).astype(np.float32)
    r.set_mesh_joints(mesh, pal)
    got = r.read_primitive_positions(mesh, 0)
    want = numpy_skin(geo.pos, joints, weights, pal)
    assert got.shape == (1024 * 1024, 3) and np.abs(got - want).max() < 5e-6
    r.close()
