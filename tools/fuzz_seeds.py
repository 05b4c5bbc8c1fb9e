"""tools/fuzz_parity.py for a list of seeds:  python tools/fuzz_seeds.py FRAMES SEED [SEED ...]   (FUZZ_SMOOTH as there)"""
import sys

sys.path.insert(0, __file__.rsplit("/tools/", 1)[0])
from oracle import orc  # noqa: E402
from renderer_amd.api import load_hip_capi  # noqa: E402
from tools import fuzz_parity as F  # noqa: E402

frames = int(sys.argv[1])
o, h = orc.capi(), load_hip_capi()
for seed in [int(v) for v in sys.argv[2:]]:
    ok, share = F.one(seed, o, h, frames)
    print(f"seed {seed} (smooth {F.SMOOTH}): {'exact images equal' if ok else 'EXACT IMAGES DIFFER'}, {share * 1e3:.2f} per thousand pixels further apart than one step", flush=True)
