"""Diagnostic run on the GPU box: renders a scene with the HIP library and the CPU restatement,
prints parity numbers and timings, and stores both images under gpurun_out/."""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import orc  # noqa: E402
from renderer_amd import scenes  # noqa: E402
from renderer_amd.api import Renderer, default_settings, load_hip_capi  # noqa: E402
from tests.helpers import psnr  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--scene", default="small")
ap.add_argument("--w", type=int, default=640)
ap.add_argument("--h", type=int, default=360)
ap.add_argument("--msaa", type=int, default=4)
ap.add_argument("--detail", type=float, default=0.25)
ap.add_argument("--tex", type=int, default=128)
ap.add_argument("--no-oracle", action="store_true")
ap.add_argument("--frames", type=int, default=3)
ap.add_argument("--tag", default="check")
a = ap.parse_args()


def build(r):
    if a.scene == "small":
        return scenes.small_scene(r, tex_size=a.tex)
    if a.scene == "helmet":
        return scenes.helmet_like(r, tex_size=a.tex, env_res=64)
    if a.scene == "sponza":
        return scenes.sponza_like(r, tex_size=a.tex, env_res=64, detail=a.detail)
    raise SystemExit("unknown scene")


st = default_settings(msaa=a.msaa)
rh = Renderer(load_hip_capi(), a.w, a.h, st)
t = time.time(); info = build(rh); print("hip scene build %.2fs" % (time.time() - t), {k: v for k, v in info.items() if k in ("triangles", "primitives")}, flush=True)
for f in range(a.frames):
    t = time.time(); out_h, avg_h = rh.render_frame(exposure_mode=1, temporal_luminance=0.5); dt = time.time() - t
    s = rh.frame_stats()
    print("hip frame %d: wall %.2f ms | device total %.3f geometry %.3f shade %.3f bloom %.3f tonemap %.3f | inst %d binned %d pairs %d views %d" % (
        f, dt * 1e3, s.ms_total, s.ms_geometry, s.ms_shade, s.ms_bloom, s.ms_tonemap, s.triangle_instances, s.triangles_binned, s.tile_pairs, s.views), flush=True)
from PIL import Image  # noqa: E402
os.makedirs("gpurun_out", exist_ok=True)
Image.fromarray(out_h[..., :3]).save(f"gpurun_out/{a.tag}_hip.png")
if not a.no_oracle:
    ro = Renderer(orc.capi(), a.w, a.h, default_settings(msaa=a.msaa))
    t = time.time(); build(ro); print("oracle scene build %.2fs" % (time.time() - t), flush=True)
    t = time.time(); out_o, avg_o = ro.render_frame(exposure_mode=1, temporal_luminance=0.5); print("oracle frame %.2fs" % (time.time() - t), flush=True)
    Image.fromarray(out_o[..., :3]).save(f"gpurun_out/{a.tag}_orc.png")
    d = np.abs(out_o.astype(int) - out_h.astype(int))
    print("8-bit max err", d.max(), "pixels differing", (d.max(-1) > 0).sum(), "of", a.w * a.h, "PSNR %.2f" % psnr(out_o, out_h), "avg", avg_o, avg_h)
    print("depth equal", np.array_equal(ro.read_depth(), rh.read_depth()), "vis equal", np.array_equal(ro.read_visibility(), rh.read_visibility()))
    for i in range(4):
        print("cascade", i, "equal", np.array_equal(ro.read_directional_shadow(i), rh.read_directional_shadow(i)))
    ho, hh = ro.read_hdr(), rh.read_hdr()
    rel = np.abs(ho - hh) / np.maximum(np.abs(ho), 1e-3)
    print("hdr max rel err %.3g, fraction of values differing %.4f" % (rel.max(), (ho != hh).mean()))
    for which, name in ((1, "irradiance"), (2, "prefiltered")):
        eq = all(np.array_equal(ro.read_env_cube(which, l), rh.read_env_cube(which, l)) for l in range(1 if which == 1 else 10))
        print(name, "cubes equal", eq)
