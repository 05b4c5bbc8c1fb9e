import sys
import numpy as np
sys.path.insert(0, ".")
from oracle import orc
from renderer_amd.api import load_hip_capi
from tests import helpers as hp
oc, hc = orc.capi(), load_hip_capi()
for (W, H) in [(250, 141), (640, 360), (1920, 1080)]:
    ro, rh = hp.make_pair(oc, hc, W, H)
    hdr = hp.random_hdr(W, H, seed=W)
    ro.post_process(hdr, 1.5); rh.post_process(hdr, 1.5)
    for level in range(6):
        bo, bh = ro.read_bloom(level), rh.read_bloom(level)
        d = np.any(bo != bh, axis=-1) if bo.ndim == 3 else (bo != bh)
        ys, xs = np.nonzero(d)
        print(W, H, "mip", level, bo.shape, "mismatch", int(d.sum()), "x", (xs.min(), xs.max()) if len(xs) else None, "y", (ys.min(), ys.max()) if len(ys) else None)
        if len(xs):
            print("   first", list(zip(xs[:6].tolist(), ys[:6].tolist())), bo[ys[0], xs[0]], bh[ys[0], xs[0]])
