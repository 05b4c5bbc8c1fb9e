"""What differs between the HIP library and the CPU checker for one seed of tools/fuzz_parity.py.
  python tools/fuzz_diag.py SEED"""
import sys

import numpy as np

sys.path.insert(0, __file__.rsplit("/tools/", 1)[0])
from tools import fuzz_parity as F  # noqa: E402
from oracle import orc  # noqa: E402
from renderer_amd.api import Renderer, default_settings, load_hip_capi  # noqa: E402

seed = int(sys.argv[1])
W, H, msaa, st, frame_kw = F.draw_settings(seed)
print(seed, W, H, st, frame_kw)
ro = Renderer(orc.capi(), W, H, default_settings(**st), keep_intermediates=True)
rh = Renderer(load_hip_capi(), W, H, default_settings(**st), keep_intermediates=True)
F.build(ro, seed); F.build(rh, seed)
a, la = ro.render_frame(**frame_kw)
b, lb = rh.render_frame(**frame_kw)
ho, hh = ro.read_hdr().astype(np.float64), rh.read_hdr().astype(np.float64)
print("avg log luminance", la, lb)
print("hdr non-finite: checker", int((~np.isfinite(ho)).sum()), "hip", int((~np.isfinite(hh)).sum()))
fin = np.isfinite(ho) & np.isfinite(hh)
for name, img, other in (("checker", ho, hh), ("hip", hh, ho)):
    yn, xn = np.nonzero(~np.isfinite(img).all(-1))
    for y, x in list(zip(yn, xn))[:4]:
        print(f"  {name} non-finite at ({x},{y}): {img[y, x]}, the other side has {other[y, x]}, triangles {ro.read_visibility()[:, y, x].tolist()}")
rel = np.abs(ho - hh) / (np.abs(ho) + 1e-3)
bad = (rel > 5e-3) & fin
print("hdr finite values apart by more than 0.5 %:", int(bad.sum()), "of", int(fin.sum()))
ys, xs, cs = np.nonzero(bad | ~fin)
vis = ro.read_visibility()
for k in range(min(8, ys.size)):
    y, x, c = ys[k], xs[k], cs[k]
    print(f"  pixel ({x},{y}) channel {c}: checker {ho[y, x]}, hip {hh[y, x]}, visibility {vis[:, y, x]}")
for level in range(6):
    bo, bh = ro.read_bloom(level), rh.read_bloom(level)
    print("bloom", level, "equal" if np.array_equal(bo, bh, equal_nan=True) else f"differs in {int((bo != bh).sum())} of {bo.size}",
          "non-finite", int((~np.isfinite(bo)).sum()), int((~np.isfinite(bh)).sum()))
d = np.abs(a.astype(int) - b.astype(int)).max(-1)
print("8-bit frame: pixels apart by more than one step:", int((d > 1).sum()), "of", d.size)
a1, la1 = ro.render_frame(**frame_kw)
b1, lb1 = rh.render_frame(**frame_kw)
print("second frame (nothing moved): checker equals its first:", np.array_equal(a, a1), "hip equals its first:", np.array_equal(b, b1), "luminance", la1, lb1)
ho1, hh1 = ro.read_hdr().astype(np.float64), rh.read_hdr().astype(np.float64)
print("  hdr: checker equals its first:", np.array_equal(ho, ho1, equal_nan=True), "hip equals its first:", np.array_equal(hh, hh1, equal_nan=True))
d1 = np.abs(b.astype(int) - b1.astype(int)).max(-1)
print("  hip frames apart:", int((d1 > 0).sum()), "pixels, max", int(d1.max()), "exposure", ro.frame_stats().exposure, rh.frame_stats().exposure)
for level in range(6):
    bo, bh = ro.read_bloom(level), rh.read_bloom(level)
    print("  bloom", level, "equal" if np.array_equal(bo, bh, equal_nan=True) else f"differs in {int((bo != bh).sum())} of {bo.size}")
