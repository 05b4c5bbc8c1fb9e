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

# is it the shadow look-up? The pixel's world position from its depth, the reference depth pbr.frag:139-143 forms for
# light `bad_light` (edit to the light the list above marks), and the four texels of that light's cube face around the look-up
if len(ys):
    from tests import glsl_f64 as G
    for r, st in ((ro, sto),):
        bench.frame_sponza(r, st, k)
    t = (k % 240) / 240.0
    cam_pos = (-11.0 + 6.0 * t, 2.2 + 0.3 * np.sin(6.283 * t), 0.6 - 0.8 * t)
    cam_eul = (-0.05, -1.45 + 0.25 * np.sin(6.283 * t), 0.0)
    vp = G.camera_view_proj(cam_pos, cam_eul, 0.1, 60.0, float(np.float32(np.radians(40.0))), float(np.float32(W) / np.float32(H)))
    z = float(dep[0, y, x])
    ndc = np.array([(x + 0.5 + (0.375 - 0.5)) / W * 2 - 1, (y + 0.5 + (0.125 - 0.5)) / H * 2 - 1, z, 1.0])   # sample 0 of the 4x pattern
    wp = np.linalg.inv(vp) @ ndc
    P = wp[:3] / wp[3]
    bad_light = 10
    lpos = np.array(lights[bad_light][0])
    L = lpos - P
    ld = np.abs(L).max() - 0.05
    ref = (100.0 - 100.0 * 0.1 / ld) / (100.0 - 0.1)
    for r in (ro, rh):
        r.set_lights(lights, scenes.default_directional_light())
    ro.render_frame(exposure_mode=1, temporal_luminance=0.5)
    casters = sorted([i for i, l in enumerate(lights) if l[2]], key=lambda i: np.linalg.norm(np.array(lights[i][0]) - np.array(cam_pos)))
    slot = casters.index(bad_light)
    face, s_, t_ = G.cube_face_coords((-L)[None, :])
    res = 512
    u, v = float(s_[0]) * res - 0.5, float(t_[0]) * res - 0.5
    i0, j0 = int(np.floor(u)), int(np.floor(v))
    fmap = ro.read_point_shadow(slot, int(face[0]))
    print(f"world position {np.round(P, 4)}, light {bad_light} at {lpos}: distance along the major axis {np.abs(L).max():.5f}, reference depth {ref:.8f}")
    print(f"  cube slot {slot} face {int(face[0])} texels around ({u:.2f},{v:.2f}):")
    for jj in (j0, j0 + 1):
        row = [float(fmap[min(max(jj, 0), res - 1), min(max(ii, 0), res - 1)]) for ii in (i0, i0 + 1)]
        print("   ", [f"{d_:.8f} ({'lit' if ref <= d_ else 'shadowed'}, eye distance {0.1 * 100.0 / (100.0 - d_ * 99.9):.5f})" for d_ in row])

# the neighbourhood of the worst pixel under `bad_light` alone, both sides (red channel of the colour image)
if len(ys):
    for r in (ro, rh):
        r.set_lights([lights[bad_light]], scenes.default_directional_light())
    ro.render_frame(exposure_mode=1, temporal_luminance=0.5)
    rh.render_frame(exposure_mode=1, temporal_luminance=0.5)
    wo, wh = ro.read_hdr()[y - 4:y + 5, x - 5:x + 6, 0], rh.read_hdr()[y - 4:y + 5, x - 5:x + 6, 0]
    np.set_printoptions(linewidth=200, precision=3, suppress=True)
    print("checker:\n", wo)
    print("library:\n", wh)

# the same with the light's shadow switched off: is it the look-up or the BRDF?
if len(ys):
    nl = (lights[bad_light][0], lights[bad_light][1], False)
    for r in (ro, rh):
        r.set_lights([nl], scenes.default_directional_light())
    ro.render_frame(exposure_mode=1, temporal_luminance=0.5)
    rh.render_frame(exposure_mode=1, temporal_luminance=0.5)
    wo, wh = ro.read_hdr()[y - 4:y + 5, x - 5:x + 6, 0], rh.read_hdr()[y - 4:y + 5, x - 5:x + 6, 0]
    print("without its shadow map, checker:\n", wo)
    print("without its shadow map, library:\n", wh)
