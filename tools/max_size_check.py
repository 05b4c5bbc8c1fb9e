"""Frames and shadow maps far larger than the benchmark's, HIP library against the CPU checker: the hall at reduced
detail with its sixteen point lights, one or two frames per size. Depth, coverage, the cascades and the point-light
faces in use bit for bit; every pixel of the 8-bit frame within one step.
  python tools/max_size_check.py WIDTH HEIGHT MSAA [DIR_RES [POINT_RES [FRAMES]]]"""
import sys
import time

import numpy as np

sys.path.insert(0, __file__.rsplit("/tools/", 1)[0])
from oracle import orc  # noqa: E402
from renderer_amd import scenes  # noqa: E402
from renderer_amd.api import Renderer, default_settings, load_hip_capi  # noqa: E402
from tests import helpers as hp  # noqa: E402

W, H, msaa = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
dir_res = int(sys.argv[4]) if len(sys.argv) > 4 else 2048
point_res = int(sys.argv[5]) if len(sys.argv) > 5 else 512
frames = int(sys.argv[6]) if len(sys.argv) > 6 else 2
st = default_settings(msaa=msaa, directional_shadow_map_resolution=dir_res, point_shadow_map_resolution=point_res)
ro = Renderer(orc.capi(), W, H, st)
rh = Renderer(load_hip_capi(), W, H, st)
infos = [scenes.sponza_like(r, tex_size=256, detail=0.5) for r in (ro, rh)]
bad = 0
for k in range(frames):
    t = k / max(frames, 2)
    for r, info in zip((ro, rh), infos):
        r.set_camera(position=(-11.0 + 9.0 * t, 2.2, 0.6 - 0.8 * t), euler=(-0.05 - 0.3 * t, -1.45 + 0.9 * t, 0.0))
        sphere, box = info["dynamic"]
        r.set_mesh_transform(sphere, scenes.translation(-3.0 + 0.5 * k, 1.2, 0.3).T)
    t0 = time.perf_counter()
    a, la = ro.render_frame(exposure_mode=1, temporal_luminance=0.5)
    t1 = time.perf_counter()
    b, lb = rh.render_frame(exposure_mode=1, temporal_luminance=0.5)
    t2 = time.perf_counter()
    print(f"frame {k}: checker {t1 - t0:.1f} s, library {(t2 - t1) * 1e3:.1f} ms (first frame includes allocation)", flush=True)
    problems = []
    x, y = ro.read_depth(), rh.read_depth()
    if not np.array_equal(x, y):
        problems.append(f"depth ({int((x != y).sum())} samples)")
    del x, y
    x, y = ro.read_visibility(), rh.read_visibility()
    if not np.array_equal(x, y):
        problems.append(f"coverage ({int((x != y).sum())} samples)")
    del x, y
    for c in range(4):
        if not np.array_equal(ro.read_directional_shadow(c), rh.read_directional_shadow(c)):
            problems.append(f"cascade {c}")
    for slot in range(8):
        for face in range(6):
            if not np.array_equal(ro.read_point_shadow(slot, face), rh.read_point_shadow(slot, face)):
                problems.append(f"point {slot}/{face}")
    worst, beyond = 0, 0
    for r0 in range(0, H, 1024):  # in bands: the difference image of a 16384^2 frame would be 8 GB of int64
        d = np.abs(a[r0:r0 + 1024].astype(np.int16) - b[r0:r0 + 1024].astype(np.int16))
        worst = max(worst, int(d.max()))
        beyond += int((d.max(-1) > 1).sum())
    if worst > 1:
        problems.append(f"8-bit frame: largest step {worst}, {beyond} pixels beyond one")
    if abs(la - lb) > 1e-4 * max(1.0, abs(la)):
        problems.append(f"luminance {la} / {lb}")
    s = rh.frame_stats()
    print(f"{W}x{H} msaa{msaa} dir {dir_res} point {point_res} frame {k}: {'equal' if not problems else 'DIFFERS: ' + ', '.join(problems)} "
          f"({s.triangles_binned} triangles binned, {s.tile_pairs} tile pairs)", flush=True)
    bad += 1 if problems else 0
ro.close(); rh.close()
sys.exit(1 if bad else 0)
