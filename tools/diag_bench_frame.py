"""Pixels of one frame of the benchmark's path where the HIP library and the CPU checker are further apart than one step.
  python tools/diag_bench_frame.py K"""
import sys

import numpy as np

sys.path.insert(0, __file__.rsplit("/tools/", 1)[0])
import bench  # noqa: E402
from oracle import orc  # noqa: E402
from renderer_amd import scenes  # noqa: E402
from renderer_amd.api import Renderer, default_settings, load_hip_capi  # noqa: E402

k = int(sys.argv[1])
W, H = 3840, 2160
ro = Renderer(orc.capi(), W, H, default_settings(msaa=4), keep_intermediates=True)
rh = Renderer(load_hip_capi(), W, H, default_settings(msaa=4), keep_intermediates=True)
sto, sth = scenes.sponza_like(ro), scenes.sponza_like(rh)
for r, st in ((ro, sto), (rh, sth)):
    bench.frame_sponza(r, st, k)
a, _ = ro.render_frame(exposure_mode=1, temporal_luminance=0.5)
b, _ = rh.render_frame(exposure_mode=1, temporal_luminance=0.5)
ho, hh = ro.read_hdr(), rh.read_hdr()
vis, dep = ro.read_visibility(), ro.read_depth()
d = np.abs(a.astype(int) - b.astype(int)).max(-1)
ys, xs = np.nonzero(d > 1)
print(len(ys), "pixels")
for y, x in zip(ys[:30], xs[:30]):
    print(f"({x},{y}) step {d[y, x]} 8-bit {a[y, x]} / {b[y, x]}  hdr {np.round(ho[y, x], 4)} / {np.round(hh[y, x], 4)}  triangles {vis[:, y, x]}  depth {dep[:, y, x]}")

# which light: the same frame with one point light at a time, then with none
if len(ys):
    y, x = ys[np.argmax(d[ys, xs])], xs[np.argmax(d[ys, xs])]
    lrng = np.random.default_rng(3)
    lights = []
    for j in range(4):
        for i in range(4):
            lights.append(((-11.25 + i * 7.5, 1.15, -5.25 + j * 3.5), tuple(20.0 + 60.0 * lrng.random(3)), (i + j) % 2 == 0))
    print(f"pixel ({x},{y}):")
    for sel in [None] + list(range(16)):
        ls = [] if sel is None else [lights[sel]]
        for r in (ro, rh):
            r.set_lights(ls, scenes.default_directional_light())
        ro.render_frame(exposure_mode=1, temporal_luminance=0.5)
        rh.render_frame(exposure_mode=1, temporal_luminance=0.5)
        po, ph = ro.read_hdr()[y, x], rh.read_hdr()[y, x]
        print("  no point light" if sel is None else f"  light {sel} casts {lights[sel][2]}", np.round(po[:3], 4), "/", np.round(ph[:3], 4), "" if np.allclose(po, ph, rtol=5e-3, atol=1e-3) else "  <-- apart")
