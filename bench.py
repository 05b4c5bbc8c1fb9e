#!/usr/bin/env python3
"""Benchmarks of the per-frame pixel pipeline on MI355X.

  python bench.py --gpus N --steps K --warmup W [--config C]

Default (config 2, the headline): frames/s on the Sponza-class scene at 3840x2160. One step = one full frame (point +
cascaded shadow maps, PBR + IBL shading, sky, blended pass, bloom chain, tonemap, luminance) of the seeded synthetic
stand-in for "Sponza 3840x2160, directional shadow + 16 point lights, IBL, bloom+tonemap" (BASELINE.json configs[1]);
the real glTF asset does not exist offline. The camera moves down the hall and the two dynamic props move every frame, so
every timed frame redraws the four cascades and the point-light cube faces the props are seen from, as the reference does
(Renderer/VulkanRendererShadow.cpp:344-388, 424); `props_at_rest` repeats the measurement with the props standing
still, where the library leaves those faces alone. The scene is resident in HBM before the timed region and the
finished RGBA8 frame stays on the device. With N > 1 every rank renders its own K frames of the camera path
(frame-batch sharding, weak scaling) and the finished images are gathered to rank 0 over RCCL, overlapped with the
next frame.

--config 1, 3, 4, 5 time the other configurations BASELINE.json names, with the same JSON shape:
  1  helmet-class object (69k triangles, 2048^2 textures) at 1920x1080: one step = one frame
  3  skinned animation batch at 1920x1080 (hall at half detail + two skinned figures re-posed every frame,
     frame k -> rank k mod N): one step = one frame
  4  environment precompute (4096x2048 panorama -> 1024^2 cube, irradiance, GGX prefilter, BRDF table): one step =
     one whole precompute; value counts device time of its kernels
  5  10-million-triangle grid at 3840x2160, orbit views: one step = one view

Prints ONE JSON line on rank 0. `roofline` describes the dominant kernel of the configuration; for the headline that
is the fragment shading kernel, bound by vector FP32 throughput (nothing in this path is a dense contraction, so the
matrix cores are unused); `roofline_passes` adds the post-process chain against the HBM roof.
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
L2_PEAK_GBS = 34500.0        # aggregate L2 bandwidth of the eight dies
# Algorithmic bytes of the post chain per pixel of the frame, by the reference's own pass structure and formats: copy of
# the colour image into bloom mip 0 (12), five down-samples (6.66), five up-samples (11.99), tonemap (20).
POST_BYTES_PER_PX = 50.65
# Bytes the fused chain of this library has to move: first down-sample reads the colour image (8) and writes mip 1 (1);
# mip 1 -> 2 (1.25); the fused small levels down (0.33) and up (0.66); mip 2 -> 1 read-modify-write (2.25); the tonemap
# pass reads colour (8) and mip 1 (1) and writes the frame (4). Bloom mip 0 and the log-luminance image outside the
# luminance window are not stored (nothing reads them).
POST_BYTES_MOVED_PER_PX = 26.5
# cube look-ups of one environment precompute: 144 per irradiance texel (6 x 256^2), 128 per prefiltered texel (6 x 349 525)
IBL_TAPS = 6 * 256 * 256 * 144 + 6 * 349525 * 128
SETUP_BYTES_PER_INSTANCE = 150.0   # triangle set-up: 12 B indices + ~7 B positions in, 128 B record out (rounded up)

METRICS = {
    1: "frames/sec DamagedHelmet-class PBR+shadow+IBL+bloom @1080p",
    2: "frames/sec Sponza-class PBR+shadow+IBL+bloom @4K",
    3: "frames/sec skinned animation batch @1080p, frame-sharded",
    4: "IBL precomputes/sec (4K equirect -> cube, irradiance, GGX prefilter, BRDF LUT), device time",
    5: "views/sec 10M-triangle mesh @4K (raster/binning stress)",
}


def shade_flops_per_px(n_point_lights):
    return 90.0 * (1 + n_point_lights) + 250.0


def measured_traffic(kernel):
    """HBM bytes per launch of `kernel` (2 x FETCH_SIZE + WRITE_SIZE) from the PMC passes committed under profiles/
    (tools/profile_gpu.sh + tools/pmc_summary.py, the headline bench command); None when no summary is there."""
    import glob
    import re
    best = None
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*pmc_summary.txt"))):
        # (a templated kernel is listed with its arguments: "k_shade_frags<false>:")
        block = re.search(r"^" + re.escape(kernel) + r"(?:<[^>\n]*>)?:.*?(?=^\S)", open(f).read() + "\nend", re.S | re.M)
        if not block:
            continue
        m_f = re.search(r"FETCH_SIZE\s+n=\d+\s+mean=([0-9.e+]+)", block.group(0))
        m_w = re.search(r"WRITE_SIZE\s+n=\d+\s+mean=([0-9.e+]+)", block.group(0))
        if m_f and m_w:
            # FETCH_SIZE counts one 128-byte line fetch as 64 bytes on gfx950, for streamed and gathered reads alike
            # (profiles/r03_fetch_size_calibration.txt, tools/fetch_calib.hip); WRITE_SIZE counts the bytes
            best = {"bytes": (2.0 * float(m_f.group(1)) + float(m_w.group(1))) * 1024.0, "source": os.path.basename(f),
                    "fetch_bytes": 2.0 * float(m_f.group(1)) * 1024.0, "write_bytes": float(m_w.group(1)) * 1024.0,
                    "note": "2 x FETCH_SIZE + WRITE_SIZE, KiB -> bytes (gfx950 tallies a 128-byte line fetch as 64 bytes; calibrated on this "
                            "kernel's access shapes, profiles/r03_fetch_size_calibration.txt)"}
    return best


def measured_valu(kernel):
    """Measured VALU figures of `kernel` from the newest profiles/*valu.json (written by tools/valu_summary.py from the
    PMC passes of the headline bench command): wave-instructions per launch and the share of SIMD cycles that issued one."""
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


# ------------------------------------------------------------------ workloads: build(renderer, args) -> state, set_frame(renderer, state, k)


def build_sponza(r, a):
    from renderer_amd import scenes
    info = scenes.sponza_like(r, tex_size=a.tex, detail=a.detail)
    if a.diag_lights >= 0:
        lrng = np.random.default_rng(3)
        lights = []
        for j in range(4):
            for i in range(4):
                lights.append(((-11.25 + i * 7.5, 1.15, -5.25 + j * 3.5), tuple(20.0 + 60.0 * lrng.random(3)), (i + j) % 2 == 0))
        r.set_lights(lights[:min(a.diag_lights, 16)], scenes.default_directional_light())
    info["n_lights"] = 16 if a.diag_lights < 0 else min(a.diag_lights, 16)
    info["what"] = (f"sponza_like {a.width}x{a.height} msaa{a.msaa}, {info['triangles']} triangles / {info['primitives']} primitives, "
                    f"{a.tex}^2 textures, dir light 4x2048^2 cascades + 16 point lights (8 shadowed, 512^2 cubes), IBL, bloom+tonemap; "
                    "camera and both dynamic props move every frame")
    return info


def frame_sponza(r, st, k, props_move=True):
    """Camera path down the hall; the sphere bobs and the box turns (frame k of the job)."""
    from renderer_amd import scenes
    t = (k % 240) / 240.0
    r.set_camera(position=(-11.0 + 6.0 * t, 2.2 + 0.3 * np.sin(6.283 * t), 0.6 - 0.8 * t), euler=(-0.05, -1.45 + 0.25 * np.sin(6.283 * t), 0.0))
    if props_move:
        sphere, box = st["dynamic"]
        r.set_mesh_transform(sphere, scenes.translation(-3.0 + 0.5 * np.sin(0.21 * k), 1.2 + 0.3 * np.sin(0.13 * k), 0.3 * np.cos(0.21 * k)).T)
        r.set_mesh_transform(box, (scenes.translation(3.0, 0.9, 0.5) @ scenes.rotation_y(0.5 + 0.05 * k)).T)


def build_helmet(r, a):
    from renderer_amd import scenes
    info = scenes.helmet_like(r, tex_size=2048)
    info["n_lights"] = 0
    info["what"] = (f"helmet_like {a.width}x{a.height} msaa{a.msaa}, {info['triangles']} triangles / 1 primitive, five 2048^2 textures, "
                    "dir light 4x2048^2 cascades, IBL, bloom+tonemap; camera orbits the object")
    return info


def frame_helmet(r, st, k):
    ang = 0.05 * k
    r.set_camera(position=(3.2 * np.sin(ang), 0.3, 3.2 * np.cos(ang)), euler=(-0.08, ang, 0.0))


def build_anim(r, a):
    from renderer_amd import scenes
    info = scenes.sponza_like(r, tex_size=512, detail=0.5)
    figs = [scenes.SkinnedFigure(r, n_bones=16, length=3.2, radius=0.3, nu=64, nv=512, tex_size=256, origin=o, seed=61 + i)
            for i, o in enumerate(((-6.0, 0.0, 0.5), (-3.0, 0.0, -1.0)))]
    r.set_camera(position=(-11.0, 2.2, 0.6), euler=(-0.05, -1.45, 0.0))
    info["figs"] = figs
    info["n_lights"] = 16
    n_skin = int(sum(p[0].pos.shape[0] for f in figs for p in f.parts))
    info["what"] = (f"sponza_like at half detail {a.width}x{a.height} msaa{a.msaa} + two skinned figures ({n_skin} skinned vertices, 16 bones each), "
                    "re-posed every frame at t = k / 30 s (CesiumMan / Fox are not on disk)")
    return info


def frame_anim(r, st, k):
    for f in st["figs"]:
        f.pose(k / 30.0)


def build_stress(r, a):
    from renderer_amd import scenes
    info = scenes.raster_stress(r, quads_per_side=a.quads)
    info["n_lights"] = 0
    info["what"] = f"raster_stress {a.width}x{a.height} msaa{a.msaa}, {info['triangles']} triangles in one primitive, dir light 4 cascades, 8 orbit views"
    return info


def frame_stress(r, st, k):
    from renderer_amd import scenes
    scenes.orbit_camera(r, k % 8)


WORKLOADS = {1: (build_helmet, frame_helmet, (1920, 1080)), 2: (build_sponza, frame_sponza, (3840, 2160)),
             3: (build_anim, frame_anim, (1920, 1080)), 5: (build_stress, frame_stress, (3840, 2160))}


def precompute_bench(a, rank, world, local_rank, dist):
    """Config 4: one step = rdr_set_environment_equirect (panorama -> cube, both bakes) + rdr_generate_brdf_lut. The API
    takes the panorama from host memory; `value` counts the device time of the kernels (rdr_get_precompute_ms), the
    wall-clock rate including the 134 MB upload is reported beside it."""
    import torch
    from renderer_amd.api import Renderer, default_settings, load_hip_capi
    r = Renderer(load_hip_capi(), 64, 64, default_settings(msaa=1), device=local_rank)
    rng = np.random.default_rng(4)
    pano = rng.random((2048, 4096, 4), dtype=np.float32) * 2.0
    dev, wall = [], []
    for k in range(a.warmup + a.steps):
        if k == a.warmup:
            torch.cuda.synchronize()
            if dist:
                dist.barrier()
            t_start = time.perf_counter()
        t0 = time.perf_counter()
        r.set_environment_equirect(pano, 1024)
        r.generate_brdf_lut(400, 1024)
        if k >= a.warmup:
            wall.append(time.perf_counter() - t0)
            dev.append(r.precompute_ms())
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    elapsed = time.perf_counter() - t_start
    if dist:
        mine = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        parts = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(parts, mine)
        elapsed = max(float(p[0]) for p in parts)
    if rank != 0:
        return None
    mean = {k: float(np.mean([d[k] for d in dev])) for k in dev[0]}
    dev_ms = sum(mean.values())
    bake_ms = mean["irradiance"] + mean["prefilter"]
    taps_per_s = IBL_TAPS / (bake_ms * 1e-3)
    cpu_baseline = None
    if world == 1 and not a.no_cpu_baseline:
        from oracle import orc
        ro = Renderer(orc.capi(), 64, 64, default_settings(msaa=1))
        tc = time.perf_counter()
        ro.set_environment_equirect(pano, 1024)
        ro.generate_brdf_lut(400, 1024)
        dt = time.perf_counter() - tc
        cpu_baseline = {"value": 1.0 / dt, "unit": "precomputes/s", "cores": int(os.environ.get("OMP_NUM_THREADS", str(os.cpu_count() or 1))), "kind": "port",
                        "sample": "one whole precompute of the same panorama, OpenMP CPU restatement (not lavapipe), wall-clock"}
        ro.close()
    return {
        "metric": METRICS[4], "value": world * 1e3 / dev_ms, "unit": "precomputes/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": dev_ms, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "4096x2048 RGBA32F panorama -> 6 x 1024^2 faces, 6 x 256^2 irradiance, 512^2 x 10 GGX-prefiltered levels, 400^2 BRDF table (1024 samples)",
                   "parallelism": f"replicas x{world}" if world > 1 else "single GPU", "lavapipe_oracle": "not available; CPU restatement used",
                   "note": "value = 1 / device time of the precompute's kernels; the call takes the panorama from host memory"},
        "roofline": {"kernel": "k_bake_irradiance + k_bake_prefilter", "bound": "l2", "achieved": taps_per_s * 16.0 / 1e9, "peak": L2_PEAK_GBS, "unit": "GB/s",
                     "frac": taps_per_s * 16.0 / 1e9 / L2_PEAK_GBS, "traffic": None, "ms": bake_ms,
                     "algorithmic": f"{IBL_TAPS} cube look-ups x 16 B (four packed texels each), cube levels resident in L2; {taps_per_s:.3e} look-ups/s"},
        "device_ms": mean,
        "wall_ms_per_step_with_upload": float(np.mean(wall)) * 1e3, "wall_precomputes_per_s": a.steps * world / elapsed,
        "cpu_baseline": cpu_baseline,
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--warmup", type=int, default=None)
    ap.add_argument("--config", type=int, default=2, choices=[1, 2, 3, 4, 5], help="which BASELINE.json configuration (default 2: the headline)")
    ap.add_argument("--width", type=int, default=None)
    ap.add_argument("--height", type=int, default=None)
    ap.add_argument("--msaa", type=int, default=4)
    ap.add_argument("--tex", type=int, default=1024)
    ap.add_argument("--detail", type=float, default=1.0)
    ap.add_argument("--quads", type=int, default=2236, help="config 5: quads per side of the grid (2236 -> 10 M triangles)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-rest", action="store_true", help="skip the second measurement of the headline (props at rest): profiling runs")
    ap.add_argument("--cpu-frames", type=int, default=None)
    ap.add_argument("--diag-lights", type=int, default=-1, help="diagnostic only: keep the first N point lights (not the benchmark workload)")
    a = ap.parse_args()
    defaults = {1: (40, 5, 5), 2: (40, 5, 5), 3: (64, 6, 3), 4: (5, 1, 1), 5: (16, 2, 1)}[a.config]   # steps, warm-up, CPU frames
    a.steps = defaults[0] if a.steps is None else a.steps
    a.warmup = defaults[1] if a.warmup is None else a.warmup
    a.cpu_frames = defaults[2] if a.cpu_frames is None else a.cpu_frames

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

    if a.config == 4:
        out = precompute_bench(a, rank, world, local_rank, dist)
        if rank == 0:
            print(json.dumps(out), flush=True)
        if dist:
            dist.destroy_process_group()
        return

    from renderer_amd.api import Renderer, default_settings, load_hip_capi

    build, set_frame, (W0, H0) = WORKLOADS[a.config]
    a.width = W0 if a.width is None else a.width
    a.height = H0 if a.height is None else a.height
    W, H = a.width, a.height
    r = Renderer(load_hip_capi(), W, H, default_settings(msaa=a.msaa), device=local_rank)
    t0 = time.time()
    info = build(r, a)
    build_s = time.time() - t0
    n_lights = info["n_lights"]

    # finished frames stay on the device; four staging slots, so that the gather of frame k has the three frames behind
    # it to finish (a communication kernel enqueued beside the shading kernel gets few wave slots and runs many times
    # longer than alone; the staging tensors are 33 MB each, rank 0 holds world x 4 more to receive into)
    from renderer_amd.sharding import FrameSharder
    sharder = FrameSharder(r, dist=dist, device=f"cuda:{local_rank}", depth=4 if world > 1 else 3)
    frame_params = dict(exposure_mode=1, temporal_luminance=0.5)
    keys = ("shade_kernel", "raster_kernel", "geometry", "shade", "bloom", "tonemap", "total", "host")

    def timed_run(first, n, frame_fn):
        """Renders job frames first .. first + n*world - 1 (n per rank) between two barriers; returns (seconds, summed pass times, last stats)."""
        acc = dict.fromkeys(keys, 0.0)
        last = [None]
        extra = {"views_reused": 0, "views": 0}

        def after_render(_k):
            s = r.frame_stats()
            last[0] = s
            acc["shade_kernel"] += s.ms_shade_kernel; acc["raster_kernel"] += s.ms_raster_kernel; acc["geometry"] += s.ms_geometry
            acc["shade"] += s.ms_shade; acc["bloom"] += s.ms_bloom; acc["tonemap"] += s.ms_tonemap; acc["total"] += s.ms_total; acc["host"] += s.host_ms
            extra["views_reused"] += int(s.views_reused); extra["views"] += int(s.views)

        torch.cuda.synchronize()
        if dist:
            dist.barrier()
        torch.cuda.synchronize()
        before = sharder.frames_rendered
        t_start = time.perf_counter()
        sharder.run(n * world, lambda rr, k: frame_fn(rr, info, first + k), frame_params, after_render=after_render)
        torch.cuda.synchronize()
        if dist:
            dist.barrier()
        torch.cuda.synchronize()
        return time.perf_counter() - t_start, acc, last[0], extra, sharder.frames_rendered - before

    # every rank renders `warmup` then `steps` frames of the job: frame k of the job goes to rank k mod world
    sharder.run(a.warmup * world, lambda rr, k: set_frame(rr, info, k), frame_params)
    elapsed, acc, last_stats, extra, rendered = timed_run(a.warmup * world, a.steps, set_frame)
    per_rank = [[elapsed, float(rendered), sharder.gather_wait_s]]
    if dist:
        mine = torch.tensor(per_rank[0], dtype=torch.float64, device="cuda")
        parts = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(parts, mine)
        per_rank = [p.tolist() for p in parts]
        elapsed = max(p[0] for p in per_rank)
    if len(per_rank) != a.gpus or any(p[1] != a.steps for p in per_rank):
        raise SystemExit(f"expected {a.gpus} ranks with {a.steps} timed frames each, saw {[(p[1]) for p in per_rank]}")
    # the same frames with the dynamic props standing still (headline configuration only): what the shadow-view reuse buys
    at_rest = None
    if a.config == 2 and not a.no_rest:
        rest_fn = lambda rr, st, k: frame_sponza(rr, st, k, props_move=False)  # noqa: E731
        sharder.run(3 * world, lambda rr, k: rest_fn(rr, info, k), frame_params)
        e2, acc2, _, extra2, _ = timed_run((a.warmup + a.steps) * world, a.steps, rest_fn)
        if dist:
            mine = torch.tensor([e2], dtype=torch.float64, device="cuda")
            parts = [torch.empty_like(mine) for _ in range(world)]
            dist.all_gather(parts, mine)
            e2 = max(float(p[0]) for p in parts)
        at_rest = {"value": a.steps * world / e2, "unit": "frames/s", "ms_per_step": e2 * 1e3 / a.steps,
                   "views_per_frame": extra2["views"] / a.steps, "views_reused_per_frame": extra2["views_reused"] / a.steps,
                   "ms_shade_kernel": acc2["shade_kernel"] / a.steps}

    cpu_baseline = None
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        # CPU restatement of the same frames on the host cores (reported baseline, not the target)
        from oracle import orc
        cores = int(os.environ.get("OMP_NUM_THREADS", str(os.cpu_count() or 1)))
        ro = Renderer(orc.capi(), W, H, default_settings(msaa=a.msaa))
        info_o = build(ro, a)
        set_frame(ro, info_o, 0)
        ro.render_frame(exposure_mode=1, temporal_luminance=0.5)  # warm-up (renders the static point-shadow maps too)
        times, passes = [], []
        for k in range(a.cpu_frames):
            set_frame(ro, info_o, 1 + k)
            tc = time.perf_counter()
            ro.render_frame(exposure_mode=1, temporal_luminance=0.5)
            times.append(time.perf_counter() - tc)
            so = ro.frame_stats()
            passes.append((so.ms_geometry, so.ms_shade, so.ms_bloom))
        mid = int(np.argsort(times)[len(times) // 2])  # the median frame
        cpu_baseline = {"value": 1.0 / float(times[mid]), "unit": "frames/s" if a.config != 5 else "views/s", "cores": cores, "kind": "port",
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
        geometry_ms = acc["geometry"] / K
        shade_flop = shade_flops_per_px(n_lights) * n_px
        shade_tf = shade_flop / (shade_ms * 1e-3) / 1e12 if shade_ms > 0 else 0.0
        post_gbs = POST_BYTES_PER_PX * n_px / (post_ms * 1e-3) / 1e9 if post_ms > 0 else 0.0
        shade_roof = {"kernel": "k_shade_frags", "bound": "valu_fp32", "achieved": shade_tf, "peak": VALU_FP32_PEAK_TF, "unit": "TFLOP/s",
                      "frac": shade_tf / VALU_FP32_PEAK_TF, "traffic": measured_traffic("k_shade_frags") if a.config == 2 else None, "ms": shade_ms,
                      "algorithmic": f"{shade_flops_per_px(n_lights):.0f} flop/px x {n_px} px" + ("" if a.config in (2, 3) else " (upper bound: the object covers part of the frame)"),
                      "valu_measured": measured_valu("k_shade_frags") if a.config == 2 else None}
        post_roof = {"pass": "bloom chain + tonemap + luminance", "bound": "hbm", "achieved": post_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": post_gbs / HBM_PEAK_GBS, "ms": post_ms, "algorithmic": f"{POST_BYTES_PER_PX} B/px x {n_px} px",
                     "moved": f"{POST_BYTES_MOVED_PER_PX} B/px by this library's fused passes",
                     "frac_by_moved_bytes": post_gbs / HBM_PEAK_GBS * POST_BYTES_MOVED_PER_PX / POST_BYTES_PER_PX}
        if a.config == 5:
            # the geometry phase dominates: priced against the HBM floor of triangle set-up (indices + positions in, record out)
            inst = float(last_stats.triangle_instances)
            gbs = SETUP_BYTES_PER_INSTANCE * inst / (geometry_ms * 1e-3) / 1e9 if geometry_ms > 0 else 0.0
            roofline = {"kernel": "geometry phase (k_chunk_cull, k_setup, scans, k_emit, sort, k_raster)", "bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS,
                        "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS, "traffic": None, "ms": geometry_ms,
                        "algorithmic": f"{SETUP_BYTES_PER_INSTANCE:.0f} B x {int(inst)} triangle instances of the last view (set-up's floor; the phase is latency- and issue-bound)"}
        elif a.config == 1 and post_ms > shade_ms:
            roofline = dict(post_roof, kernel="post chain (k_bloom_*, k_tonemap, k_lum_reduce)", traffic=None)
        else:
            roofline = shade_roof
        out = {
            "metric": METRICS[a.config],
            "value": K * world / elapsed,
            "unit": "frames/s" if a.config != 5 else "views/s",
            "n_gpus": world,
            "steps": K,
            "warmup": a.warmup,
            "ms_per_step": ms_step,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": info["what"],
                       "parallelism": f"frame-sharded x{world}, RCCL gather of RGBA8 frames to rank 0" if world > 1 else "single GPU",
                       "lavapipe_oracle": "not available; CPU restatement used"},
            "roofline": roofline,
            "roofline_passes": [post_roof],
            "device_ms": {k: v / K for k, v in acc.items()},
            "frame_counters": {"triangle_instances": int(last_stats.triangle_instances), "triangles_binned": int(last_stats.triangles_binned),
                               "tile_pairs": int(last_stats.tile_pairs), "views": int(last_stats.views), "views_reused": int(last_stats.views_reused),
                               "views_per_frame": extra["views"] / K, "views_reused_per_frame": extra["views_reused"] / K},
            "props_at_rest": at_rest,
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
