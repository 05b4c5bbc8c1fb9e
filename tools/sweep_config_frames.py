"""The other configurations' scenes at their stated sizes, HIP library against the CPU checker, over several frames:
  helmet   the helmet-class sphere, 1920x1080 4x, from random directions and distances
  anim     the hall at half detail with two skinned figures re-posed every frame (frames k of the benchmark's job), 1920x1080 4x
  prims    ten thousand one-primitive meshes (the draw lists are built on the GPU), 1920x1080 4x, random cameras
Depth, coverage, the cascades and the point-light faces in use bit for bit; every pixel of the 8-bit frame within one step.
  python tools/sweep_config_frames.py helmet|anim|prims [frames]"""
import sys
from types import SimpleNamespace

import numpy as np

sys.path.insert(0, __file__.rsplit("/tools/", 1)[0])
import bench  # noqa: E402
from oracle import orc  # noqa: E402
from renderer_amd import scenes  # noqa: E402
from renderer_amd.api import Renderer, default_settings, load_hip_capi  # noqa: E402
from tests import helpers as hp  # noqa: E402

which = sys.argv[1]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 8
W, H = 1920, 1080
ro = Renderer(orc.capi(), W, H, default_settings(msaa=4), keep_intermediates=True)
rh = Renderer(load_hip_capi(), W, H, default_settings(msaa=4), keep_intermediates=True)
rng = np.random.default_rng(11)
n_cubes = 0
if which == "helmet":
    for r in (ro, rh):
        scenes.helmet_like(r)
elif which == "anim":
    a = SimpleNamespace(width=W, height=H, msaa=4)
    so_, sh_ = bench.build_anim(ro, a), bench.build_anim(rh, a)
    n_cubes = 8
else:
    for r in (ro, rh):
        scenes.many_primitives(r)
    n_cubes = 8
bad = 0
for i in range(n):
    if which == "helmet":
        d = rng.normal(0, 1, 3); d /= np.linalg.norm(d)
        dist = float(rng.uniform(1.3, 6.0))
        pos = d * dist
        yaw = float(np.arctan2(pos[0], pos[2]))
        pitch = float(-np.arcsin(pos[1] / dist))
        what = f"from {tuple(round(float(v), 2) for v in pos)}"
        for r in (ro, rh):
            r.set_camera(position=tuple(pos), euler=(pitch, yaw, 0.0))
    elif which == "anim":
        k = int(rng.integers(0, 256))
        what = f"frame {k}"
        for r, st in ((ro, so_), (rh, sh_)):
            bench.frame_anim(r, st, k)
            t = k / 256.0
            r.set_camera(position=(-11.0 + 6.0 * t, 2.2, 0.6 - 0.8 * t), euler=(-0.05, -1.45 + 0.2 * np.sin(6.283 * t), 0.0))
    else:
        pos = (float(rng.uniform(-40, 40)), float(rng.uniform(1.0, 30.0)), float(rng.uniform(-40, 40)))
        eul = (float(rng.uniform(-1.2, 0.1)), float(rng.uniform(-3.1, 3.1)), 0.0)
        what = f"camera at {tuple(round(v, 1) for v in pos)}"
        for r in (ro, rh):
            r.set_camera(position=pos, euler=eul)
    a_, la = ro.render_frame(exposure_mode=1, temporal_luminance=0.5)
    b_, lb = rh.render_frame(exposure_mode=1, temporal_luminance=0.5)
    problems = []
    if not np.array_equal(ro.read_depth(), rh.read_depth()):
        problems.append("depth")
    if not np.array_equal(ro.read_visibility(), rh.read_visibility()):
        problems.append("coverage")
    for c in range(4):
        if not np.array_equal(ro.read_directional_shadow(c), rh.read_directional_shadow(c)):
            problems.append(f"cascade {c}")
    for slot in range(n_cubes):
        for face in range(6):
            try:
                x, y = ro.read_point_shadow(slot, face), rh.read_point_shadow(slot, face)
            except Exception:
                continue
            if not np.array_equal(x, y):
                problems.append(f"point {slot}/{face}")
    d = np.abs(a_.astype(int) - b_.astype(int))
    psnr = hp.psnr(a_, b_)
    if d.max() > 1 or psnr < 45.0:
        problems.append(f"8-bit frame: largest step {int(d.max())}, {int((d.max(-1) > 1).sum())} pixels beyond one, {psnr:.1f} dB")
    print(f"{which} {what}: {'equal' if not problems else 'DIFFERS: ' + ', '.join(problems)} ({psnr:.1f} dB, {ro.frame_stats().triangles_binned} triangles binned)", flush=True)
    bad += 1 if problems else 0
print(f"{which}: {n} frames, {bad} with a difference")
sys.exit(1 if bad else 0)
