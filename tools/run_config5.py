"""BASELINE config 5 at full size on the HIP library: a 10-million-triangle displaced grid (one primitive) seen from
8 orbit views at 3840x2160, 4x MSAA, directional light with 4 cascades. Prints one JSON line: frames/s over the 8
views, set-up/raster/shade times and the frame counters; checks that coverage is consistent with depth."""
import json
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from renderer_amd import scenes  # noqa: E402
from renderer_amd.api import Renderer, default_settings, load_hip_capi  # noqa: E402

quads = int(sys.argv[1]) if len(sys.argv) > 1 else 2236
W, H = 3840, 2160
t0 = time.perf_counter()
r = Renderer(load_hip_capi(), W, H, default_settings(msaa=4))
info = scenes.raster_stress(r, quads_per_side=quads)
build_s = time.perf_counter() - t0
scenes.orbit_camera(r, 0)
r.render_frame(exposure_mode=1, temporal_luminance=0.5)      # warm-up: buffers reach their sizes
times, stats = [], []
for k in range(8):
    scenes.orbit_camera(r, k)
    t = time.perf_counter()
    out, _ = r.render_frame(exposure_mode=1, temporal_luminance=0.5)
    times.append(time.perf_counter() - t)
    s = r.frame_stats()
    stats.append({"ms_geometry": round(s.ms_geometry, 3), "ms_shade": round(s.ms_shade, 3), "ms_total": round(s.ms_total, 3),
                  "instances": int(s.triangle_instances), "binned": int(s.triangles_binned), "pairs": int(s.tile_pairs)})
depth, vis = r.read_depth(), r.read_visibility()
ok = bool(np.array_equal(vis == 0xFFFFFFFF, depth >= 1.0)) and float(depth.min()) >= 0.0
print(json.dumps({"triangles": info["triangles"], "views": 8, "frames_per_s_host_clock": round(8 / sum(times), 2),
                  "device_ms_per_view": round(float(np.mean([s["ms_total"] for s in stats])), 3), "scene_build_s": round(build_s, 1),
                  "coverage_consistent_with_depth": ok, "nonblank": bool(out[..., :3].std() > 5), "per_view": stats}))
