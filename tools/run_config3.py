"""BASELINE config 3 on one GPU: an animation batch at 1920x1080 (4x MSAA) with skinned, key-framed figures in
the hall scene, frames t_k = k / 30 s sharded through FrameSharder (frame k -> rank k mod world; here world = 1 unless
launched under torchrun). The named assets (CesiumMan / Fox) are not on disk: two procedural limbs of 16 bones and
~33k vertices each stand in. Prints one JSON line."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, ".")
import torch  # noqa: E402

from renderer_amd import scenes  # noqa: E402
from renderer_amd.api import Renderer, default_settings, load_hip_capi  # noqa: E402
from renderer_amd.sharding import FrameSharder  # noqa: E402

frames = int(sys.argv[1]) if len(sys.argv) > 1 else 256
rank, world, local = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1)), int(os.environ.get("LOCAL_RANK", 0))
dist = None
torch.cuda.set_device(local)
if world > 1:
    import torch.distributed as dist
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
W, H = 1920, 1080
r = Renderer(load_hip_capi(), W, H, default_settings(msaa=4), device=local)
scenes.sponza_like(r, tex_size=512, detail=0.5)
figs = [scenes.SkinnedFigure(r, n_bones=16, length=3.2, radius=0.3, nu=64, nv=512, tex_size=256, origin=o, seed=61 + i)
        for i, o in enumerate(((-6.0, 0.0, 0.5), (-3.0, 0.0, -1.0)))]
r.set_camera(position=(-11.0, 2.2, 0.6), euler=(-0.05, -1.45, 0.0))


def set_frame(rr, k):
    for f in figs:
        f.pose(k / 30.0)


sharder = FrameSharder(r, dist=dist, device=f"cuda:{local}", depth=2)
params = dict(exposure_mode=1, temporal_luminance=0.5)
sharder.run(4 * world, set_frame, params)
torch.cuda.synchronize()
t0 = time.perf_counter()
sharder.run(frames, set_frame, params)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
s = r.frame_stats()
if rank == 0:
    print(json.dumps({"frames": frames, "world": world, "frames_per_s": round(frames / dt, 1), "ms_per_frame_device": round(s.ms_total, 3),
                      "skinned_vertices": int(sum(p[0].pos.shape[0] for f in figs for p in f.parts)), "resolution": [W, H]}))
