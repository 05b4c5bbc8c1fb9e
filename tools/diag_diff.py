"""Diagnostic (GPU box): renders the reduced config-2 scene with both libraries and describes the worst pixels."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import orc
from renderer_amd import scenes
from renderer_amd.api import Renderer, default_settings, load_hip_capi

kw = dict(msaa=4, directional_shadow_map_resolution=1024, point_shadow_map_resolution=256)
ro = Renderer(orc.capi(), 640, 360, default_settings(**kw)); rh = Renderer(load_hip_capi(), 640, 360, default_settings(**kw))
for r in (ro, rh):
    scenes.sponza_like(r, tex_size=128, env_res=32, detail=0.25)
for frame in range(2):
    if frame == 1:
        for r in (ro, rh):
            r.set_camera(position=(-6.0, 2.5, 0.2), euler=(-0.05, -1.3, 0.0))
    oo, _ = ro.render_frame(exposure_mode=1, temporal_luminance=0.5); oh, _ = rh.render_frame(exposure_mode=1, temporal_luminance=0.5)
    d = np.abs(oo.astype(int) - oh.astype(int)).max(-1)
    print("frame", frame, "max", d.max(), "n>1", (d > 1).sum())
    ho, hh = ro.read_hdr(), rh.read_hdr()
    vis = rh.read_visibility(); dep = rh.read_depth()
    ys, xs = np.nonzero(d > 1)
    for y, x in list(zip(ys, xs))[:12]:
        print("  px", x, y, "out", oo[y, x], oh[y, x], "hdr", ho[y, x], hh[y, x], "vis", vis[:, y, x], "depth", dep[:, y, x])
    print("  depth eq", np.array_equal(ro.read_depth(), rh.read_depth()), "vis eq", np.array_equal(ro.read_visibility(), vis))
    for s in range(8):
        for f in range(6):
            if not np.array_equal(ro.read_point_shadow(s, f), rh.read_point_shadow(s, f)): print("  point shadow differs", s, f)
