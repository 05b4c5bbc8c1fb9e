"""The C++ façade (include/renderer.hpp: the reference's Camera / Node / Mesh / light classes and the public
surface of its renderer class over the C ABI), exercised by a host program written the way Renderer/main.cpp is.

CPU: tests/cpp/facade_scene.cpp is compiled against the CPU checker library and its frames are compared with the
same scene built through the Python API. GPU: the binary linked against the HIP library (built by
__graft_entry__.build()) must match the checker-linked one to the parity bar."""
import os
import subprocess

import numpy as np
import pytest

from renderer_amd import _binding as B
from renderer_amd import animation as A
from renderer_amd import scenes
from renderer_amd.api import Renderer, default_settings, quat_look_at
from tests import helpers as hp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CPP = os.path.join(ROOT, "tests", "cpp")
W, H = 96, 64


def build_orc_binary():
    exe = os.path.join(CPP, "facade_scene_orc")
    src = os.path.join(CPP, "facade_scene.cpp")
    hdr = os.path.join(ROOT, "include", "renderer.hpp")
    if not os.path.exists(exe) or os.path.getmtime(exe) < max(os.path.getmtime(src), os.path.getmtime(hdr)):
        from oracle import orc
        orc.build()
        subprocess.run(["g++", "-std=c++17", "-O1", "-Wall", "-DRDR_ABI_PREFIX=orc_", src, "-o", exe, "-L" + os.path.join(ROOT, "oracle"),
                        "-lorc", "-Wl,-rpath,$ORIGIN/../../oracle", "-lpthread"], check=True)
    return exe


def run_binary(exe, tmp_path, tag):
    out = tmp_path / f"{tag}.raw"
    p = subprocess.run([exe, str(out)], check=True, capture_output=True, text=True, timeout=120)
    frames = np.fromfile(out, np.uint8).reshape(3, H, W, 4)
    return frames, p.stdout


def quat_from_euler(e):
    cx, cy, cz = np.cos(np.float32(e) * np.float32(0.5))
    sx, sy, sz = np.sin(np.float32(e) * np.float32(0.5))
    return np.array([cx * cy * cz + sx * sy * sz, sx * cy * cz - cx * sy * sz, cx * sy * cz + sx * cy * sz, cx * cy * sz - sx * sy * cz], np.float64)


def trs(t, q, s):
    w, x, y, z = q
    r = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
                  [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                  [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])
    m = np.eye(4)
    m[:3, :3] = r * np.asarray(s)[None, :]
    m[:3, 3] = t
    return m


def python_frames(capi):
    """The scene of facade_scene.cpp through the Python API (row-major matrices here, transposed on upload)."""
    r = Renderer(capi, W, H, default_settings(msaa=4, directional_shadow_map_resolution=128, point_shadow_map_resolution=32))
    tex = np.zeros((16, 16, 4), np.uint8)
    for y in range(16):
        for x in range(16):
            on = ((x // 4) + (y // 4)) & 1
            tex[y, x] = (230, 200, 150, 255) if on else (60, 80, 200, 255)
    checker = r.make_texture(r.load_image(tex, B.FORMAT_R8G8B8A8_SRGB))
    ident = [1.0, 0.0, 0.0, 0.0]

    def plate_model(arm_t):
        root = trs((0.0, 0.2, -0.5), ident, (1.5, 1.5, 1.5))
        arm = trs(arm_t, quat_from_euler((0.0, 0.4, 0.0)), (1, 1, 1))
        plate = trs((0.0, 0.6, 0.0), quat_from_euler((0.5, 0.0, 0.0)), (1, 1, 1))
        return root @ arm @ plate

    m_floor = scenes.make_material([checker, B.INVALID_ID, B.INVALID_ID, B.INVALID_ID, B.INVALID_ID], metallic=0.0, roughness=0.7)
    m_plate = scenes.make_material(base_color=(0.9, 0.3, 0.2, 1.0), emissive=(0.0, 0.0, 0.2), metallic=0.8, roughness=0.3)
    r.add_mesh([scenes.upload_primitive(r, scenes.plane(8.0, 8.0, 4, 4, uv_scale=4.0), m_floor, B.UINT16)])
    plate = r.add_mesh([scenes.upload_primitive(r, scenes.plane(1.2, 1.2, 2, 2), m_plate, B.UINT16)], model=plate_model((0.5, 0.0, 0.0)).T,
                       is_static=False)
    # the skinned flap: two bones, joints as uint8, float weights
    flap = scenes.plane(1.0, 1.0, 4, 4)
    v = np.float32(0.5) - flap.pos[:, 2]
    w1 = np.clip(np.float32(2.0) * v - np.float32(0.5), 0, 1).astype(np.float32)
    joints = np.tile(np.uint8([0, 1, 0, 0]), (v.size, 1))
    weights = np.stack([np.float32(1.0) - w1, w1, np.zeros_like(w1), np.zeros_like(w1)], 1)
    bone1 = A.Node(translation=(0, 0, -0.5))
    bone0 = A.Node(translation=(0, 0, 0.5), children=[bone1])
    flap_node = A.Node(translation=(-1.5, 0.8, 0.5), children=[bone0])
    bone1.rotation_animation = A.Animation([0.0, 1.0], [A.quat_from_axis_angle((1, 0, 0), 0.0), A.quat_from_axis_angle((1, 0, 0), 0.9)],
                                           A.LINEAR, A.MIRROR, rotation=True)
    bone0.translation_animation = A.Animation([0.0, 2.0], [[0, 0, 0], [0, 0, 0.5], [0, 1, 0], [0, 0, 0], [0, 0.4, 0.5], [0, 0, 0]], A.CUBICSPLINE, A.CLAMP)
    flap_node.recalculate_model()
    skin = A.Skin([bone0, bone1], [scenes.translation(0, 0, -0.5), np.eye(4)])
    m_flap = scenes.make_material([checker, B.INVALID_ID, B.INVALID_ID, B.INVALID_ID, B.INVALID_ID], metallic=0.1, roughness=0.5)
    flap_mesh = r.add_mesh([scenes.upload_skinned_primitive(r, flap, joints, weights, m_flap, B.UINT8, B.FLOAT)], model=flap_node.model.T, is_static=False)
    env = np.zeros((6, 8, 8, 4), np.float32)
    for f in range(6):
        for y in range(8):
            for x in range(8):
                env[f, y, x] = (np.float32(0.2) + np.float32(0.1) * np.float32(f), np.float32(0.3) + np.float32(0.05) * np.float32(y),
                                np.float32(0.6) + np.float32(0.02) * np.float32(x), 1.0)
    r.set_environment_map(env)
    lut = np.zeros((16, 16, 3), np.uint8)
    lut[..., 0] = (np.arange(16) * 16)[None, :]
    lut[..., 1] = (np.arange(16) * 8)[:, None]
    r.set_brdf_lut(lut)
    cam_pos, cam_euler = np.array([0.2, 1.6, 4.0]), (-0.25, 0.05, 0.0)
    r.set_camera(position=cam_pos, euler=cam_euler)
    d = np.array([0.05, -1.0, 0.05], np.float32)
    d /= np.float32(np.linalg.norm(d))
    r.set_lights([((1.0, 2.0, 1.0), (12, 10, 8), True), ((-2.0, 1.5, 0.5), (3, 4, 6), False)],
                 ((0, 100, 0), (10, 7, 2), quat_look_at(d, (1.0, 0.0, 0.0))))
    f0, _ = r.render_frame(exposure_mode=1, temporal_luminance=0.5)
    f0 = f0.copy()
    r.set_mesh_transform(plate, plate_model((-0.4, 0.1, 0.3)).T)
    # Camera::dolly(0.5): 0.5 along the camera's -z axis; view = Rx(-ex) Ry(-ey) Rz(-ez) T(-p)
    ex, ey = cam_euler[0], cam_euler[1]
    rx = np.array([[1, 0, 0], [0, np.cos(-ex), -np.sin(-ex)], [0, np.sin(-ex), np.cos(-ex)]])
    ry = np.array([[np.cos(-ey), 0, np.sin(-ey)], [0, 1, 0], [-np.sin(-ey), 0, np.cos(-ey)]])
    cam_pos = cam_pos + np.linalg.inv(rx @ ry) @ np.array([0.0, 0.0, -0.5])
    r.set_camera(position=cam_pos, euler=cam_euler)
    f1, _ = r.render_frame(exposure_mode=1, temporal_luminance=0.5)
    f1 = f1.copy()
    flap_node.set_animation_time(1.4)
    flap_node.recalculate_model()
    r.set_mesh_joints(flap_mesh, skin.palette_for_abi(flap_node.model))
    f2, _ = r.render_frame(exposure_mode=1, temporal_luminance=0.5)
    return np.stack([f0, f1, f2]), cam_pos, bone1.model[:3, 3]


def assert_frames_close(a, b):
    for k in range(3):
        d = np.abs(a[k].astype(int) - b[k].astype(int)).max(-1)
        # host matrices are built in float here and in double in the Python twin: a silhouette pixel may flip
        assert (d > 1).mean() < 2e-3, f"frame {k}: {(d > 1).sum()} pixels differ by more than 1/255"
        assert hp.psnr(a[k], b[k]) >= 45.0


def test_facade_program_matches_python_api_on_the_checker(orc_capi, tmp_path):
    frames, stdout = run_binary(build_orc_binary(), tmp_path, "orc")
    assert "plate model col3 = -0.600000 1.250000 -0.050000" in stdout  # root T/S, arm T/R, plate T by hand
    ref, cam_pos, bone1_pos = python_frames(orc_capi)
    bone_line = [ln for ln in stdout.splitlines() if ln.startswith("bone1")][0]
    assert np.allclose([float(v) for v in bone_line.split("=")[1].split()], bone1_pos, atol=1e-5), "same spline and hierarchy on both sides"
    assert not np.array_equal(frames[1], frames[2]), "the flap moved"
    cam_line = [ln for ln in stdout.splitlines() if ln.startswith("camera")][0]
    assert np.allclose([float(v) for v in cam_line.split("=")[1].split()], cam_pos, atol=1e-5)
    assert not np.array_equal(frames[0], frames[1])
    assert frames[..., :3].std() > 10, "the frames must not be blank"
    assert_frames_close(frames, ref)


@pytest.mark.gpu
def test_facade_program_on_the_hip_library(hip_capi, tmp_path):
    exe = os.path.join(CPP, "facade_scene_hip")
    assert os.path.exists(exe), "tests/cpp/facade_scene_hip is built by __graft_entry__.build()"
    hip_frames, _ = run_binary(exe, tmp_path, "hip")
    orc_frames, _ = run_binary(build_orc_binary(), tmp_path, "orc")
    for k in range(3):
        hp.assert_image_close(orc_frames[k], hip_frames[k])


@pytest.mark.gpu
def test_sharded_host_program_gathers_over_rccl(hip_capi, tmp_path):
    """tests/cpp/shard_host.cpp: a C++ host that renders a frame batch rank by rank and brings the frames to rank 0 with
    the grouped ncclSend / ncclRecv of include/rdr_gather.h, started as a fresh process per rank. One GPU here, so the
    world is one rank: the communicator, the send-to-self group, the pipelined loop (frame k + 1 enqueued before frame
    k is gathered) and the luminance all-gather all run; the frames must equal those of the same program rendering
    one frame at a time without the gather library."""
    exe = os.path.join(CPP, "shard_host_hip")
    assert os.path.exists(exe), "tests/cpp/shard_host_hip is built by __graft_entry__.build()"
    n = 7
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    serial = subprocess.run([exe, str(tmp_path / "serial.raw"), str(n), "0", "0", str(tmp_path / "unused")], check=True, capture_output=True,
                            text=True, timeout=300, env=env)
    shard = subprocess.run([exe, str(tmp_path / "shard.raw"), str(n), "1", "0", str(tmp_path / "id"), "0"], check=True, capture_output=True,
                           text=True, timeout=300, env=env)
    a = np.fromfile(tmp_path / "serial.raw", np.uint8).reshape(n, 96, 160, 4)
    b = np.fromfile(tmp_path / "shard.raw", np.uint8).reshape(n, 96, 160, 4)
    for k in range(n):
        assert np.array_equal(a[k], b[k]), f"frame {k}"
    assert len({a[k].tobytes() for k in range(n)}) == n and a[..., :3].std() > 10
    lum = lambda text: [ln for ln in text.splitlines() if ln.startswith("luminance")]  # noqa: E731  (RCCL prints a banner of its own)
    assert lum(serial.stdout) == lum(shard.stdout) and len(lum(serial.stdout)) == n
