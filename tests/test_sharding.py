"""Frame-batch sharding (renderer_amd/sharding.py) with two processes over gloo, on the CPU.

The ranks drive the CPU checker library through the same C ABI the HIP library exports, so the partition,
the asynchronous gather with staging-slot reuse, the padding of the last round and the luminance exchange are
exercised exactly as bench.py uses them with nccl (RCCL)."""
import os
import socket
import sys

import numpy as np
import pytest

from renderer_amd import sharding

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
W, H, N_FRAMES = 64, 36, 5


def _set_frame(r, k):
    r.set_camera(position=(0.4 + 0.3 * k, 1.7, 5.2 - 0.2 * k), euler=(-0.2, 0.05 + 0.04 * k, 0.0))


def _make_renderer():
    from oracle import orc
    from renderer_amd import scenes
    from renderer_amd.api import Renderer, default_settings
    r = Renderer(orc.capi(), W, H, default_settings(msaa=1, directional_shadow_map_resolution=128, point_shadow_map_resolution=32))
    scenes.small_scene(r, tex_size=16, env_res=8)
    return r


def _worker(rank, world, port, out_dir, pipelined=False):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    torch.set_num_threads(1)
    os.environ["OMP_NUM_THREADS"] = "2"
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    try:
        r = _make_renderer()
        sh = sharding.FrameSharder(r, dist=dist, device="cpu", depth=2, pipelined=pipelined)
        got = {}
        stats_seen = []

        def on_frame(k, img):
            assert k not in got
            got[k] = img.numpy().copy()

        def after_render(k):
            stats_seen.append((k, int(r.frame_stats().frame_index)))

        lum = sh.run(N_FRAMES, _set_frame, dict(exposure_mode=1, temporal_luminance=0.5), on_frame, after_render=after_render)
        # per-frame statistics are collected for the right frame: the j-th frame this rank rendered is its context's frame j + 1
        assert stats_seen == [(k, j + 1) for j, k in enumerate(sharding.frames_of_rank(N_FRAMES, rank, world))]
        assert sh.frames_rendered == len(sharding.frames_of_rank(N_FRAMES, rank, world)) and sh.gather_wait_s >= 0.0
        assert sorted(lum) == sharding.frames_of_rank(N_FRAMES, rank, world)
        table = sh.gather_luminance(N_FRAMES)
        np.save(os.path.join(out_dir, f"lum_{rank}.npy"), table)
        if rank == 0:
            assert sorted(got) == list(range(N_FRAMES))
            np.save(os.path.join(out_dir, "frames.npy"), np.stack([got[k] for k in range(N_FRAMES)]))
        dist.barrier()
    finally:
        dist.destroy_process_group()


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_partition():
    assert sharding.frames_of_rank(10, 1, 4) == [1, 5, 9]
    assert sharding.frames_of_rank(3, 3, 4) == []
    assert sharding.rounds(10, 4) == 3 and sharding.rounds(8, 4) == 2 and sharding.rounds(0, 4) == 0
    allk = sorted(k for rk in range(8) for k in sharding.frames_of_rank(256, rk, 8))
    assert allk == list(range(256))
    assert all(len(sharding.frames_of_rank(256, rk, 8)) == 32 for rk in range(8))
    with pytest.raises(ValueError):
        sharding.frames_of_rank(4, 4, 4)


def test_two_rank_gloo_gather_matches_serial(tmp_path):
    import torch.multiprocessing as mp
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    frames = np.load(tmp_path / "frames.npy")
    # serial rendering of the same job in this process
    r = _make_renderer()
    sh = sharding.FrameSharder(r, dist=None, device="cpu", depth=2)
    serial = {}
    lum = sh.run(N_FRAMES, _set_frame, dict(exposure_mode=1, temporal_luminance=0.5), lambda k, img: serial.__setitem__(k, img.numpy().copy()))
    assert sorted(serial) == list(range(N_FRAMES))
    for k in range(N_FRAMES):
        assert np.array_equal(frames[k], serial[k]), f"frame {k} differs between the 2-rank and the serial run"
    assert len({frames[k].tobytes() for k in range(N_FRAMES)}) == N_FRAMES, "frames must differ from each other"
    t0, t1 = np.load(tmp_path / "lum_0.npy"), np.load(tmp_path / "lum_1.npy")
    assert np.array_equal(t0, t1)
    assert np.allclose(t0, [lum[k] for k in range(N_FRAMES)], atol=0)


def test_two_rank_gloo_pipelined_branch_matches_serial(tmp_path):
    """The branch bench.py takes on GPUs (two frames in flight: a frame is only enqueued, its results are collected and
    its gather issued once the next frame has been enqueued behind it), driven here over gloo with the CPU checker,
    which honours the same hand-over points for the statistics. Odd frame count: the last round has one rank without
    a frame, which has to drain its pipeline before it takes part in the gather."""
    import torch.multiprocessing as mp
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path), True), nprocs=2, join=True)
    frames = np.load(tmp_path / "frames.npy")
    r = _make_renderer()
    serial = {}
    sh = sharding.FrameSharder(r, dist=None, device="cpu", depth=2, pipelined=True)
    lum = sh.run(N_FRAMES, _set_frame, dict(exposure_mode=1, temporal_luminance=0.5), lambda k, img: serial.__setitem__(k, img.numpy().copy()))
    assert sorted(serial) == list(range(N_FRAMES))
    for k in range(N_FRAMES):
        assert np.array_equal(frames[k], serial[k]), f"frame {k}"
    t0, t1 = np.load(tmp_path / "lum_0.npy"), np.load(tmp_path / "lum_1.npy")
    assert np.array_equal(t0, t1) and np.allclose(t0, [lum[k] for k in range(N_FRAMES)], atol=0)


def test_exposure_sequence_follows_the_reference_recurrence():
    """VulkanRenderer.cpp:1079-1082 by hand for two frames at dt = 1/30."""
    avg = [1.0, 2.0, 0.5]
    tl, ex = sharding.exposure_sequence(avg, 1.0 / 30.0)
    g = 1.0 - np.exp(-1.0 / 30.0)
    t0 = 0.0 + (2.0 ** 0.0 - 0.0) * g          # first frame sees luminance 0 -> 2^0
    t1 = t0 + (2.0 ** 1.0 - t0) * g
    t2 = t1 + (2.0 ** 2.0 - t1) * g
    assert np.allclose(tl, [t0, t1, t2], rtol=1e-6)
    assert np.allclose(ex, [0.18 / (1.03 - 2.0 / (np.log2(t + 1.0) + 2.0)) for t in (t0, t1, t2)], rtol=1e-5)


@pytest.mark.gpu
@pytest.mark.parametrize("in_flight", [2, 3])
@pytest.mark.parametrize("exposure_mode", [1, 0])
def test_two_frames_in_flight_match_frame_by_frame_rendering(hip_capi, exposure_mode, in_flight, monkeypatch):
    """Frames enqueued back to back into device tensors (two or, with RDR_FRAMES_IN_FLIGHT=3, three in flight; the
    sharder collects a frame once the renderer reports it complete, and has two staging tensors, so with three in flight
    it waits for a frame before reusing its tensor) against the same frames rendered to completion one at a time: same images, same
    per-frame luminance and counters, and the statistics carry the right frame index. With exposure_mode 0 every
    frame's exposure follows from the luminance of the frame before it (the reference's recurrence), which the
    renderer has to wait for in the middle of the next frame."""
    import torch

    from renderer_amd import scenes
    from renderer_amd.api import Renderer, default_settings

    W, H, n = 200, 120, 5
    monkeypatch.setenv("RDR_FRAMES_IN_FLIGHT", str(in_flight))

    def build():
        r = Renderer(hip_capi, W, H, default_settings(msaa=4, directional_shadow_map_resolution=512, point_shadow_map_resolution=128))
        scenes.small_scene(r)
        return r

    def camera(r, k):
        r.set_camera(position=(0.3 * k - 0.5, 1.2 + 0.1 * k, 4.0 - 0.2 * k), euler=(-0.1, 0.05 * k, 0.0))

    serial, serial_stats = [], []
    r1 = build()
    for k in range(n):
        camera(r1, k)
        img, avg = r1.render_frame(exposure_mode=exposure_mode, temporal_luminance=0.5, delta_time=1.0 / 30.0)
        s = r1.frame_stats()
        serial.append(img.copy())
        serial_stats.append((avg, int(s.triangles_binned), int(s.tile_pairs), int(s.frame_index)))
    r1.close()

    r2 = build()
    sh = sharding.FrameSharder(r2, dist=None, device="cuda:0", depth=2)
    piped, piped_stats = {}, {}

    def after_render(k):
        s = r2.frame_stats()
        piped_stats[k] = (float(s.avg_log_luminance), int(s.triangles_binned), int(s.tile_pairs), int(s.frame_index))
        assert s.ms_total > 0.0 and s.ms_shade_kernel > 0.0

    lum = sh.run(n, camera, dict(exposure_mode=exposure_mode, temporal_luminance=0.5, delta_time=1.0 / 30.0), on_frame=lambda k, t: piped.__setitem__(k, t.cpu().numpy().copy()),
                 after_render=after_render)
    torch.cuda.synchronize()
    r2.close()
    assert sorted(piped) == list(range(n)) and sorted(piped_stats) == list(range(n))
    for k in range(n):
        assert np.array_equal(piped[k], serial[k]), f"frame {k}"
        assert piped_stats[k][1:] == serial_stats[k][1:], f"counters of frame {k}"
        assert piped_stats[k][0] == serial_stats[k][0] == lum[k], f"luminance of frame {k}"
        assert piped_stats[k][3] == k + 1


@pytest.mark.gpu
@pytest.mark.parametrize("in_flight", [2, 3])
def test_frames_in_flight_with_changing_scene_state(hip_capi, in_flight, monkeypatch):
    """Between frames that are still in flight (two, or three with RDR_FRAMES_IN_FLIGHT=3) the caller moves the camera, a dynamic mesh and the lights, replaces
    the environment (frame 4) and the BRDF table (frame 6), adds a mesh with an image of its own (frame 5), and now and
    then asks for a frame in host memory (which drains the pipeline). Every frame must equal the same frame rendered by
    a second context one frame at a time: nothing a frame reads may be overwritten by the frame enqueued behind it."""
    import torch

    from renderer_amd import scenes
    from renderer_amd.api import Renderer, default_settings

    W, H, n = 160, 96, 9
    monkeypatch.setenv("RDR_FRAMES_IN_FLIGHT", str(in_flight))

    def build():
        r = Renderer(hip_capi, W, H, default_settings(msaa=4, directional_shadow_map_resolution=256, point_shadow_map_resolution=64))
        ids = scenes.small_scene(r)
        return r, ids

    def set_state(r, ids, k):
        r.set_camera(position=(0.25 * k - 0.8, 1.1 + 0.05 * k, 4.2 - 0.15 * k), euler=(-0.08, 0.04 * k, 0.0))
        m = scenes.translation(0.4 * np.sin(0.9 * k), 0.9 + 0.08 * k, 0.4 - 0.1 * k)
        r.set_mesh_transform(ids["sphere"], m.T)
        r.set_lights([((1.5 - 0.2 * k, 2.2, 1.0 + 0.1 * k), (20.0 + 3 * k, 18.0, 14.0), True),
                      ((-1.0, 1.5 + 0.1 * k, 2.0), (6.0, 9.0 + k, 12.0), k % 2 == 0)], scenes.default_directional_light())
        if k == 4:
            r.set_environment_map(scenes.sky_faces(16, seed=9, sun_dir=(-0.5, 0.6, 0.2)))
        if k == 5:
            tex = scenes.material_textures(r, 32, 23)
            r.add_mesh([scenes.upload_primitive(r, scenes.box(0.5, 0.5, 0.5, 1), scenes.make_material(tex), scenes.B.UINT16)],
                       model=scenes.translation(-0.9, 0.6, 0.8).T, is_static=False)
        if k == 6:
            r.set_brdf_lut(scenes.brdf_lut(24, 24))

    params = dict(exposure_mode=1, temporal_luminance=0.4)
    r1, ids1 = build()
    serial = []
    for k in range(n):
        set_state(r1, ids1, k)
        img, _ = r1.render_frame(**params)
        serial.append(img.copy())
    r1.close()

    r2, ids2 = build()
    slots = [torch.empty((H, W, 4), dtype=torch.uint8, device="cuda:0") for _ in range(n)]
    host = {}
    for k in range(n):
        set_state(r2, ids2, k)
        if k % 4 == 3:
            host[k], _ = r2.render_frame(**params)
        else:
            r2.render_frame(device_ptr=slots[k].data_ptr(), **params)
    r2.synchronize()
    assert r2.frame_stats().frame_index == n
    for k in range(n):
        got = host[k] if k in host else slots[k].cpu().numpy()
        assert np.array_equal(got, serial[k]), f"frame {k}"
    r2.close()


def _nccl_world_of_one(port, out_dir, root):
    """Child process: the branch bench.py --gpus N takes on every rank (torch.distributed "nccl" = RCCL, device staging
    tensors, frames in flight, asynchronous gathers retired when their slot comes round again), with a world of one."""
    sys.path.insert(0, root)
    import torch
    import torch.distributed as dist

    from renderer_amd import scenes
    from renderer_amd.api import Renderer, default_settings, load_hip_capi
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        Wn, Hn, n = 320, 180, 10
        r = Renderer(load_hip_capi(), Wn, Hn, default_settings(msaa=4, directional_shadow_map_resolution=512, point_shadow_map_resolution=128))
        ids = scenes.small_scene(r)

        def set_frame(rr, k):
            rr.set_camera(position=(0.3 * k - 0.5, 1.2 + 0.1 * k, 4.0 - 0.2 * k), euler=(-0.1, 0.05 * k, 0.0))
            rr.set_mesh_transform(ids["sphere"], scenes.translation(0.1 * k, 0.9, 0.4).T)

        sh = sharding.FrameSharder(r, dist=dist, device="cuda:0", depth=4)
        got = {}
        lum = sh.run(n, set_frame, dict(exposure_mode=1, temporal_luminance=0.5), on_frame=lambda k, t: got.__setitem__(k, t.cpu().numpy().copy()))
        table = sh.gather_luminance(n)
        torch.cuda.synchronize()
        assert sorted(got) == list(range(n)) and sh.frames_rendered == n
        np.save(os.path.join(out_dir, "frames.npy"), np.stack([got[k] for k in range(n)]))
        np.save(os.path.join(out_dir, "lum.npy"), np.array([lum[k] for k in range(n)], np.float32))
        np.save(os.path.join(out_dir, "table.npy"), table)
        # the same frames one at a time, in the same process
        r2 = Renderer(load_hip_capi(), Wn, Hn, default_settings(msaa=4, directional_shadow_map_resolution=512, point_shadow_map_resolution=128))
        ids2 = scenes.small_scene(r2)
        serial, serial_lum = [], []
        for k in range(n):
            r2.set_camera(position=(0.3 * k - 0.5, 1.2 + 0.1 * k, 4.0 - 0.2 * k), euler=(-0.1, 0.05 * k, 0.0))
            r2.set_mesh_transform(ids2["sphere"], scenes.translation(0.1 * k, 0.9, 0.4).T)
            img, avg = r2.render_frame(exposure_mode=1, temporal_luminance=0.5)
            serial.append(img.copy()); serial_lum.append(avg)
        np.save(os.path.join(out_dir, "serial.npy"), np.stack(serial))
        np.save(os.path.join(out_dir, "serial_lum.npy"), np.array(serial_lum, np.float32))
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
def test_rccl_gather_branch_with_a_world_of_one(hip_capi, tmp_path):
    """FrameSharder with the nccl backend and device tensors, in a fresh process (as every rank of bench.py --gpus N
    is): RCCL work handles, the retire logic that orders the renderer's streams against the gather, four staging
    slots over ten frames. Gathered frames must equal frame-by-frame rendering byte for byte."""
    import subprocess
    code = ("import sys; sys.path.insert(0, %r); from tests.test_sharding import _nccl_world_of_one; _nccl_world_of_one(%d, %r, %r)"
            % (ROOT, _free_port(), str(tmp_path), ROOT))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, env=env)
    assert p.returncode == 0, p.stderr[-3000:]
    frames, serial = np.load(tmp_path / "frames.npy"), np.load(tmp_path / "serial.npy")
    for k in range(frames.shape[0]):
        assert np.array_equal(frames[k], serial[k]), f"frame {k}"
    assert len({f.tobytes() for f in frames}) == frames.shape[0]
    assert np.array_equal(np.load(tmp_path / "lum.npy"), np.load(tmp_path / "serial_lum.npy"))
    assert np.array_equal(np.load(tmp_path / "table.npy"), np.load(tmp_path / "lum.npy"))
