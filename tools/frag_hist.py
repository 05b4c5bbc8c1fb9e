"""Distribution of shaded fragments (distinct triangles among a pixel's samples) per 16x16 tile of the benchmark frame.
Runs on the GPU box: python tools/frag_hist.py"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from renderer_amd import scenes  # noqa: E402
from renderer_amd.api import Renderer, default_settings, load_hip_capi  # noqa: E402

W, H = 3840, 2160
r = Renderer(load_hip_capi(), W, H, default_settings(msaa=4), device=0)
scenes.sponza_like(r)
bench.camera_for_frame(r, 20)
r.render_frame(exposure_mode=1, temporal_luminance=0.5)
vis = np.asarray(r.read_visibility()).reshape(4, H, W)
dep = np.asarray(r.read_depth()).reshape(4, H, W)
vis = np.where(dep >= 1.0, 0xFFFFFFFF, vis).astype(np.uint32)
s = np.sort(vis, axis=0)
n = (s[0] != 0xFFFFFFFF).astype(np.int32)
for k in range(1, 4):
    n += ((s[k] != s[k - 1]) & (s[k] != 0xFFFFFFFF)).astype(np.int32)
print("pixels", W * H, "fragments", int(n.sum()), "per pixel", n.sum() / (W * H), "hist", np.bincount(n.ravel(), minlength=5))
t = n.reshape(H // 16, 16, W // 16, 16).sum(axis=(1, 3)).ravel()
print("tiles", t.size, "mean", t.mean(), "zero", int((t == 0).sum()), "<=256", int((t <= 256).sum()), "257..320", int(((t > 256) & (t <= 320)).sum()),
      "321..512", int(((t > 320) & (t <= 512)).sum()), ">512", int((t > 512).sum()), "max", int(t.max()))
passes = np.ceil(t / 256.0)
wave_passes = (np.floor(t / 256.0) * 4 + np.ceil((t % 256) / 64.0)).sum()
print("block passes", int(passes.sum()), "wave passes now", int(wave_passes), "dense wave passes", int(np.ceil(t.sum() / 64.0)),
      "per-tile dense", int(np.ceil(t / 64.0).sum()))
q = n.reshape(H // 8, 8, W // 8, 8).sum(axis=(1, 3)).ravel()
print("quadrants: mean", q.mean(), "passes", int(np.ceil(q / 64.0).sum()))
