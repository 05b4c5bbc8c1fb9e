"""A long run of the benchmark's frames on one context: frames per second per block of frames, free device memory and the
process's resident set before and after (nothing may grow once the first frames have sized the buffers).
  python tools/soak.py [frames]"""
import resource
import sys
import time
from types import SimpleNamespace

import numpy as np

sys.path.insert(0, __file__.rsplit("/tools/", 1)[0])
import torch  # noqa: E402
import bench  # noqa: E402
from renderer_amd.api import Renderer, default_settings, load_hip_capi  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
a = SimpleNamespace(width=3840, height=2160, msaa=4, tex=1024, detail=1.0, diag_lights=-1)
r = Renderer(load_hip_capi(), a.width, a.height, default_settings(msaa=4), device=0)
info = bench.build_sponza(r, a)
slots = [torch.empty((a.height, a.width, 4), dtype=torch.uint8, device="cuda:0") for _ in range(3)]
kw = dict(exposure_mode=1, temporal_luminance=0.5)
for k in range(30):
    bench.frame_sponza(r, info, k)
    r.render_frame(device_ptr=slots[k % 3].data_ptr(), **kw)
r.synchronize()
free0, rss0 = torch.cuda.mem_get_info(0)[0], resource.getrusage(resource.RUSAGE_SELF).ru_maxrss
block = max(n // 10, 1)
t0 = time.perf_counter()
for k in range(n):
    bench.frame_sponza(r, info, 30 + k)
    r.render_frame(device_ptr=slots[k % 3].data_ptr(), **kw)
    if (k + 1) % block == 0:
        r.synchronize()
        t1 = time.perf_counter()
        print(f"frames {k + 1 - block}..{k}: {block / (t1 - t0):.1f} frames/s, free device memory {torch.cuda.mem_get_info(0)[0] / 2**20:.0f} MiB", flush=True)
        if k + 1 == block:   # the camera path has been walked once by now: buffers that follow the scene's density have their size
            free0, rss0 = torch.cuda.mem_get_info(0)[0], resource.getrusage(resource.RUSAGE_SELF).ru_maxrss
        t0 = time.perf_counter()
free1, rss1 = torch.cuda.mem_get_info(0)[0], resource.getrusage(resource.RUSAGE_SELF).ru_maxrss
print(f"after the first {block} frames device memory in use grew by {(free0 - free1) / 2**20:.1f} MiB, the peak resident set by {(rss1 - rss0) / 1024:.1f} MiB ({n} frames)")
print("last frame's image: mean", float(slots[(n - 1) % 3].float().mean()))
r.close()
sys.exit(0 if free0 - free1 < 64 * 2**20 and rss1 - rss0 < 256 * 1024 else 1)
