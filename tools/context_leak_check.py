"""Contexts opened, used (a small hall, three frames, read-backs; draw lists on the host and on the GPU in turn) and closed,
forty times: device memory and the resident set must not grow.
  python tools/context_leak_check.py"""
import os
import resource
import sys

sys.path.insert(0, __file__.rsplit("/tools/", 1)[0])
import torch
from renderer_amd import scenes
from renderer_amd.api import Renderer, default_settings, load_hip_capi
capi = load_hip_capi()
torch.cuda.init()
def once(k):
    os.environ["RDR_DEVICE_DRAW_LISTS"] = str(k % 2)
    r = Renderer(capi, 640, 360, default_settings(msaa=4 if k % 2 else 1, directional_shadow_map_resolution=512, point_shadow_map_resolution=128), device=0)
    scenes.sponza_like(r, tex_size=64, env_res=16, detail=0.2)
    for f in range(3):
        r.render_frame(exposure_mode=1, temporal_luminance=0.5)
    r.read_depth(); r.read_visibility()
    r.close()
for k in range(4): once(k)
free0, rss0 = torch.cuda.mem_get_info(0)[0], resource.getrusage(resource.RUSAGE_SELF).ru_maxrss
for k in range(40): once(k)
free1, rss1 = torch.cuda.mem_get_info(0)[0], resource.getrusage(resource.RUSAGE_SELF).ru_maxrss
print(f"40 contexts opened, used and closed: device memory in use grew by {(free0 - free1) / 2**20:.1f} MiB, peak resident set by {(rss1 - rss0) / 1024:.1f} MiB")
sys.exit(0 if free0 - free1 < 16 * 2**20 and rss1 - rss0 < 64 * 1024 else 1)
