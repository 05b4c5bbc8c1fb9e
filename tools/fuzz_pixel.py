"""One pixel of one seed of tools/fuzz_parity.py, library against checker: the whole scene, then every primitive alone
(same random draws, the others left out), at the seed's sample count and at one sample per pixel.
  python tools/fuzz_pixel.py SEED X Y"""
import sys

import numpy as np

sys.path.insert(0, __file__.rsplit("/tools/", 1)[0])
from oracle import orc  # noqa: E402
from renderer_amd.api import Renderer, default_settings, load_hip_capi  # noqa: E402
from tools import fuzz_parity as F  # noqa: E402

seed, X, Y = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
W, H, msaa, st, frame_kw = F.draw_settings(seed)
n_prims = int(np.random.default_rng(seed).integers(8, 40) if seed >= F.BIG_FROM else np.random.default_rng(seed).integers(1, 6))
print(f"seed {seed}: {W}x{H} msaa {msaa}, {n_prims} primitives, {st}")
o, h = orc.capi(), load_hip_capi()
for samples in (msaa, 1):
    for only in [None] + [{p} for p in range(n_prims)]:
        s2 = dict(st, msaa=samples)
        ro = Renderer(o, W, H, default_settings(**s2), keep_intermediates=True)
        rh = Renderer(h, W, H, default_settings(**s2), keep_intermediates=True)
        F.build(ro, seed, only); F.build(rh, seed, only)
        a, _ = ro.render_frame(**frame_kw)
        b, _ = rh.render_frame(**frame_kw)
        ho, hh = ro.read_hdr().astype(np.float64), rh.read_hdr().astype(np.float64)
        vis, dep = ro.read_visibility(), ro.read_depth()
        same = np.array_equal(vis, rh.read_visibility()) and np.array_equal(dep, rh.read_depth())
        rel = np.abs(ho[..., :3] - hh[..., :3]).max(-1) / (np.abs(ho[..., :3]).max(-1) + 1e-2)
        ys, xs = np.nonzero(rel > 0.02)
        print(f"samples {samples} primitives {'all' if only is None else sorted(only)}: exact images {'equal' if same else 'DIFFER'}; pixel ({X},{Y}) linear "
              f"{np.round(ho[Y, X, :3], 4).tolist()} / {np.round(hh[Y, X, :3], 4).tolist()} triangles {vis[:, Y, X].tolist()} depth {dep[:, Y, X].tolist()}; "
              f"{ys.size} pixels with linear colours more than 2 % apart {list(zip(xs.tolist(), ys.tolist()))[:6]}", flush=True)
        ro.close(); rh.close()
