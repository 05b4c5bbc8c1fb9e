"""The float64 known answers of tests/test_shading_kat.py (check_jittered) over a range of seeds: other cameras, lights,
frame sizes, sample counts and material variants of the textured scene.
  python tools/kat_sweep.py orc|hip FIRST_SEED END_SEED [ROUGHNESS_SCALE]
ROUGHNESS_SCALE multiplies every material's roughness factor (0.1: mirror-like surfaces, highlights a few pixels wide)."""
import sys

import numpy as np
sys.path.insert(0, __file__.rsplit("/tools/", 1)[0])
from tests import test_shading_kat as T
from _pytest.monkeypatch import MonkeyPatch
which = sys.argv[1]
if which == "orc":
    from oracle import orc
    capi = orc.capi()
else:
    from renderer_amd.api import load_hip_capi
    capi = load_hip_capi()
bad = 0
scale = float(sys.argv[4]) if len(sys.argv) > 4 else 1.0
from tests import kat_scene as K
_prims = K.prims


def scaled_prims(variant="base"):
    opaque, blended = _prims(variant)
    for p in opaque + blended:
        p.mat["roughness"] = float(np.float32(p.mat["roughness"] * scale))
    return opaque, blended


for seed in range(int(sys.argv[2]), int(sys.argv[3])):
    mp = MonkeyPatch()
    if scale != 1.0:
        mp.setattr(K, "prims", scaled_prims)
    try:
        n = T.check_jittered(capi, seed, mp)
        print(seed, "ok", n, flush=True)
    except AssertionError as e:
        bad += 1
        print(seed, "FAIL", str(e)[:300], flush=True)
    finally:
        mp.undo()
print("failures:", bad)
