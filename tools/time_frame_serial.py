"""The benchmark frame rendered one frame at a time (host destination: nothing of the next frame overlaps), for
per-kernel times without other kernels beside them. Run under tools/kstats_cmd.sh."""
import sys
import types
sys.path.insert(0, ".")
import numpy as np
from renderer_amd.api import Renderer, default_settings, load_hip_capi
import bench
W, H = 3840, 2160
a = types.SimpleNamespace(width=W, height=H, msaa=4, tex=1024, detail=1.0, diag_lights=-1)
r = Renderer(load_hip_capi(), W, H, default_settings(msaa=4))
st = bench.build_sponza(r, a)
out = np.empty((H, W, 4), np.uint8)
for k in range(8):
    bench.frame_sponza(r, st, k)
    r.render_frame(out=out, exposure_mode=1, temporal_luminance=0.5)
r.close()
