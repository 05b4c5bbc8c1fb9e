#!/usr/bin/env python3
"""Headline benchmark: frames/s of the per-frame pixel pipeline on the Sponza-class scene at 4K.

  python bench.py --gpus N --steps K --warmup W

One step = one full frame (point + cascaded shadow maps, PBR + IBL shading of the visibility buffer, sky,
blended pass, bloom chain, tonemap, luminance) of the seeded synthetic stand-in for "Sponza 3840x2160,
directional shadow + 16 point lights, IBL, bloom+tonemap" (BASELINE.json configs[1]); the real glTF asset does
not exist offline. The scene is resident in HBM before the timed region and the finished RGBA8 frame stays on
the device. With N > 1 every rank renders its own K frames of the camera path (frame-batch sharding, weak
scaling) and the finished images are gathered to rank 0 over RCCL, overlapped with the next frame.

Prints ONE JSON line on rank 0. `roofline` describes the dominant kernel (the shading kernel, bound by vector
FP32 throughput: nothing in this path is a dense contraction, so the matrix cores are unused);
`roofline_passes` adds the post-process chain against the HBM roof.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # MI355X HBM3E peak
VALU_FP32_PEAK_TF = 157.3    # MI355X vector FP32 peak
# Algorithmic bytes of the post chain per pixel of the frame, by the reference's own pass structure and formats: copy of
# the colour image into bloom mip 0 (12), five down-samples (6.66), five up-samples (11.99), tonemap (20).
POST_BYTES_PER_PX = 50.65
# Bytes the fused chain of this library has to move: first down-sample reads the colour image (8) and writes mip 1 (1);
# mip 1 -> 2 (1.25); the fused small levels down (0.33) and up (0.66); mip 2 -> 1 read-modify-write (2.25); the tonemap
# pass reads colour (8) and mip 1 (1) and writes the frame (4). Bloom mip 0 and the log-luminance image outside the
# luminance window are not stored (nothing reads them).
POST_BYTES_MOVED_PER_PX = 26.5


def shade_flops_per_px(n_point_lights):
    return 90.0 * (1 + n_point_lights) + 250.0


def measured_traffic(kernel):
    """HBM bytes per launch of `kernel` (FETCH_SIZE + WRITE_SIZE) from the PMC passes committed under profiles/
    (tools/profile_gpu.sh + tools/pmc_summary.py, the same bench command); None when no summary is there."""
    import glob
    import re
    best = None
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*pmc_summary.txt"))):
        block = re.search(r"^" + re.escape(kernel) + r":.*?(?=^\S)", open(f).read() + "\nend", re.S | re.M)
        if not block:
            continue
        m_f = re.search(r"FETCH_SIZE\s+n=\d+\s+mean=([0-9.e+]+)", block.group(0))
        m_w = re.search(r"WRITE_SIZE\s+n=\d+\s+mean=([0-9.e+]+)", block.group(0))
        if m_f and m_w:
            best = {"bytes": (float(m_f.group(1)) + float(m_w.group(1))) * 1024.0, "source": os.path.basename(f),
                    "note": "FETCH_SIZE + WRITE_SIZE, KiB -> bytes, uncorrected (4-byte gathers: uncalibrated on gfx950)"}
    return best


def measured_valu(kernel):
    """Measured VALU figures of `kernel` from the newest profiles/*valu.json (written by tools/valu_summary.py from the
    PMC passes of this same bench command): wave-instructions per launch and the share of SIMD cycles that issued one."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*valu.json")))
    if not files:
        return None
    d = json.load(open(files[-1]))
    e = d.get(kernel)
    if e:
        e = dict(e, source=os.path.basename(files[-1]))
    return e


def launch_ranks(n):
    """`--gpus N` without a launcher around it: starts N fresh ranks (one per GPU) under torch.distributed.run as a
    child process, before this process has touched a GPU, and exits with the child's status."""
    import socket
    import subprocess
    import torch
    have = torch.cuda.device_count()  # counting devices does not initialise the GPU
    if have < n:
        raise SystemExit(f"bench.py --gpus {n}: only {have} GPU(s) visible; refusing to report a {n}-GPU number from fewer")
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    raise SystemExit(subprocess.run(cmd).returncode)


def camera_for_frame(r, k):
    """Camera path down the hall; frame k of the job."""
    t = (k % 240) / 240.0
    r.set_camera(position=(-11.0 + 6.0 * t, 2.2 + 0.3 * np.sin(6.283 * t), 0.6 - 0.8 * t), euler=(-0.05, -1.45 + 0.25 * np.sin(6.283 * t), 0.0))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--width", type=int, default=3840)
    ap.add_argument("--height", type=int, default=2160)
    ap.add_argument("--msaa", type=int, default=4)
    ap.add_argument("--tex", type=int, default=1024)
    ap.add_argument("--detail", type=float, default=1.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-frames", type=int, default=5)
    ap.add_argument("--diag-lights", type=int, default=-1, help="diagnostic only: keep the first N point lights (not the benchmark workload)")
    a = ap.parse_args()

    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        launch_ranks(a.gpus)
    import torch
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}: every rank of the job must be started (one per GPU)")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP library is the only compute path")
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    from renderer_amd import scenes
    from renderer_amd.api import Renderer, default_settings, load_hip_capi

    W, H = a.width, a.height
    r = Renderer(load_hip_capi(), W, H, default_settings(msaa=a.msaa), device=local_rank)
    t0 = time.time()
    info = scenes.sponza_like(r, tex_size=a.tex, detail=a.detail)
    build_s = time.time() - t0
    n_lights = 16
    if a.diag_lights >= 0:
        lrng = np.random.default_rng(3)
        lights = []
        for j in range(4):
            for i in range(4):
                lights.append(((-11.25 + i * 7.5, 1.15, -5.25 + j * 3.5), tuple(20.0 + 60.0 * lrng.random(3)), (i + j) % 2 == 0))
        n_lights = min(a.diag_lights, 16)
        r.set_lights(lights[:n_lights], scenes.default_directional_light())

    # finished frames stay on the device; four staging slots, so that the gather of frame k has the three frames behind
    # it to finish (a communication kernel enqueued beside the shading kernel gets few wave slots and runs many times
    # longer than alone; the staging tensors are 33 MB each, rank 0 holds world x 4 more to receive into)
    from renderer_amd.sharding import FrameSharder
    sharder = FrameSharder(r, dist=dist, device=f"cuda:{local_rank}", depth=4 if world > 1 else 3)
    frame_params = dict(exposure_mode=1, temporal_luminance=0.5)
    acc = {"shade_kernel": 0.0, "raster_kernel": 0.0, "geometry": 0.0, "shade": 0.0, "bloom": 0.0, "tonemap": 0.0, "total": 0.0, "host": 0.0}
    last_stats = [None]

    def after_render(_k):
        s = r.frame_stats()
        last_stats[0] = s
        acc["shade_kernel"] += s.ms_shade_kernel; acc["raster_kernel"] += s.ms_raster_kernel; acc["geometry"] += s.ms_geometry
        acc["shade"] += s.ms_shade; acc["bloom"] += s.ms_bloom; acc["tonemap"] += s.ms_tonemap; acc["total"] += s.ms_total; acc["host"] += s.host_ms

    # every rank renders `warmup` then `steps` frames of the camera path: frame k of the job goes to rank k mod world
    sharder.run(a.warmup * world, camera_for_frame, frame_params)
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    t_start = time.perf_counter()
    sharder.run(a.steps * world, lambda rr, k: camera_for_frame(rr, a.warmup * world + k), frame_params, after_render=after_render)
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t_start
    per_rank = [[elapsed, float(sharder.frames_rendered - a.warmup), sharder.gather_wait_s]]
    if dist:
        mine = torch.tensor(per_rank[0], dtype=torch.float64, device="cuda")
        parts = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(parts, mine)
        per_rank = [p.tolist() for p in parts]
        elapsed = max(p[0] for p in per_rank)
    if len(per_rank) != a.gpus or any(p[1] != a.steps for p in per_rank):
        raise SystemExit(f"expected {a.gpus} ranks with {a.steps} timed frames each, saw {[(p[1]) for p in per_rank]}")
    last_stats = last_stats[0]

    cpu_baseline = None
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        # CPU restatement of the same frame on the host cores (reported baseline, not the target)
        from oracle import orc
        cores = int(os.environ.get("OMP_NUM_THREADS", str(os.cpu_count() or 1)))
        ro = Renderer(orc.capi(), W, H, default_settings(msaa=a.msaa))
        scenes.sponza_like(ro, tex_size=a.tex, detail=a.detail)
        camera_for_frame(ro, 0)
        ro.render_frame(exposure_mode=1, temporal_luminance=0.5)  # warm-up (renders the static point-shadow maps too)
        times, passes = [], []
        for k in range(a.cpu_frames):
            camera_for_frame(ro, 1 + k)
            tc = time.perf_counter()
            ro.render_frame(exposure_mode=1, temporal_luminance=0.5)
            times.append(time.perf_counter() - tc)
            so = ro.frame_stats()
            passes.append((so.ms_geometry, so.ms_shade, so.ms_bloom))
        mid = int(np.argsort(times)[len(times) // 2])  # the median frame
        cpu_baseline = {"value": 1.0 / float(times[mid]), "unit": "frames/s", "cores": cores, "kind": "port",
                        "sample": f"median of {a.cpu_frames} frames of the same {W}x{H} workload after 1 warm-up, OpenMP CPU restatement (not lavapipe)",
                        "ms_per_pass": {"shadow + main raster": passes[mid][0], "shading + blended pass": passes[mid][1],
                                        "bloom + tonemap + luminance": passes[mid][2]},
                        "frame_ms_all": [round(t * 1e3, 1) for t in times]}
        ro.close()

    if rank == 0:
        K = a.steps
        n_px = W * H
        ms_step = elapsed * 1e3 / K
        shade_ms = acc["shade_kernel"] / K
        post_ms = (acc["bloom"] + acc["tonemap"]) / K
        shade_flop = shade_flops_per_px(n_lights) * n_px
        shade_tf = shade_flop / (shade_ms * 1e-3) / 1e12 if shade_ms > 0 else 0.0
        post_gbs = POST_BYTES_PER_PX * n_px / (post_ms * 1e-3) / 1e9 if post_ms > 0 else 0.0
        out = {
            "metric": "frames/sec Sponza-class PBR+shadow+IBL+bloom @4K",
            "value": K * world / elapsed,
            "unit": "frames/s",
            "n_gpus": world,
            "steps": K,
            "warmup": a.warmup,
            "ms_per_step": ms_step,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": f"sponza_like {W}x{H} msaa{a.msaa}, {info['triangles']} triangles / {info['primitives']} primitives, "
                                   f"{a.tex}^2 textures, dir light 4x2048^2 cascades + 16 point lights (8 shadowed, 512^2 cubes), IBL, bloom+tonemap",
                       "parallelism": f"frame-sharded x{world}, RCCL gather of RGBA8 frames to rank 0" if world > 1 else "single GPU",
                       "lavapipe_oracle": "not available; CPU restatement used"},
            "roofline": {"kernel": "k_shade_frags", "bound": "valu_fp32", "achieved": shade_tf, "peak": VALU_FP32_PEAK_TF, "unit": "TFLOP/s",
                         "frac": shade_tf / VALU_FP32_PEAK_TF, "traffic": measured_traffic("k_shade_frags"), "ms": shade_ms,
                         "algorithmic": f"{shade_flops_per_px(n_lights):.0f} flop/px x {n_px} px", "valu_measured": measured_valu("k_shade_frags")},
            "roofline_passes": [
                {"pass": "bloom chain + tonemap + luminance", "bound": "hbm", "achieved": post_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                 "frac": post_gbs / HBM_PEAK_GBS, "ms": post_ms, "algorithmic": f"{POST_BYTES_PER_PX} B/px x {n_px} px",
                 "moved": f"{POST_BYTES_MOVED_PER_PX} B/px by this library's fused passes",
                 "frac_by_moved_bytes": post_gbs / HBM_PEAK_GBS * POST_BYTES_MOVED_PER_PX / POST_BYTES_PER_PX},
            ],
            "device_ms": {k: v / K for k, v in acc.items()},
            "frame_counters": {"triangle_instances": int(last_stats.triangle_instances), "triangles_binned": int(last_stats.triangles_binned),
                               "tile_pairs": int(last_stats.tile_pairs), "views": int(last_stats.views)},
            "ranks_seen": len(per_rank),
            "per_rank": [{"frames_per_s": p[1] / p[0], "gather_wait_ms_per_frame": p[2] * 1e3 / max(p[1] + a.warmup, 1.0)} for p in per_rank],
            "scene_build_s": build_s,
            "cpu_baseline": cpu_baseline,
        }
        print(json.dumps(out), flush=True)
    if dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
