"""The benchmark's scene at the benchmark's size (Sponza-class, 3840x2160, 4x MSAA), HIP library against the CPU checker,
along the benchmark's own camera path with its props moving and from random places in the hall: depth, coverage, the four
cascades and all 48 point-light faces bit for bit, the 8-bit frame within one step and 45 dB.
  python tools/sweep_bench_frames.py [frames]"""
import sys

import numpy as np

sys.path.insert(0, __file__.rsplit("/tools/", 1)[0])
import bench  # noqa: E402
from oracle import orc  # noqa: E402
from renderer_amd import scenes  # noqa: E402
from renderer_amd.api import Renderer, default_settings, load_hip_capi  # noqa: E402
from tests import helpers as hp  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 12
W, H = 3840, 2160
ro = Renderer(orc.capi(), W, H, default_settings(msaa=4), keep_intermediates=True)
rh = Renderer(load_hip_capi(), W, H, default_settings(msaa=4), keep_intermediates=True)
sto, sth = scenes.sponza_like(ro), scenes.sponza_like(rh)
rng = np.random.default_rng(7)
bad = 0
for i in range(n):
    if i % 2 == 0:
        k = int(rng.integers(0, 2000))
        what = f"frame {k} of the benchmark's path"
        for r, st in ((ro, sto), (rh, sth)):
            bench.frame_sponza(r, st if isinstance(st, dict) else {"dynamic": st}, k)
    else:
        pos = (float(rng.uniform(-13, 13)), float(rng.uniform(0.3, 9.0)), float(rng.uniform(-5, 5)))
        eul = (float(rng.uniform(-0.7, 0.5)), float(rng.uniform(-3.1, 3.1)), float(rng.uniform(-0.1, 0.1)))
        what = f"camera at {tuple(round(v, 2) for v in pos)} looking {tuple(round(v, 2) for v in eul)}"
        for r in (ro, rh):
            r.set_camera(position=pos, euler=eul)
    a, la = ro.render_frame(exposure_mode=1, temporal_luminance=0.5)
    b, lb = rh.render_frame(exposure_mode=1, temporal_luminance=0.5)
    problems = []
    if not np.array_equal(ro.read_depth(), rh.read_depth()):
        problems.append("depth")
    if not np.array_equal(ro.read_visibility(), rh.read_visibility()):
        problems.append("coverage")
    for c in range(4):
        if not np.array_equal(ro.read_directional_shadow(c), rh.read_directional_shadow(c)):
            problems.append(f"cascade {c}")
    for slot in range(8):
        for face in range(6):
            if not np.array_equal(ro.read_point_shadow(slot, face), rh.read_point_shadow(slot, face)):
                problems.append(f"point {slot}/{face}")
    d = np.abs(a.astype(int) - b.astype(int))
    psnr = hp.psnr(a, b)
    if d.max() > 1 or psnr < 45.0:
        problems.append(f"8-bit frame: largest step {int(d.max())}, {int((d.max(-1) > 1).sum())} pixels beyond one, {psnr:.1f} dB")
    so, sh = ro.frame_stats(), rh.frame_stats()
    print(f"{what}: {'equal' if not problems else 'DIFFERS: ' + ', '.join(problems)} ({psnr:.1f} dB, {so.triangles_binned} triangles binned, "
          f"luminance {la:.4f} / {lb:.4f})", flush=True)
    bad += 1 if problems else 0
print(f"{n} frames, {bad} with a difference")
sys.exit(1 if bad else 0)
