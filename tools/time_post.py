"""Post-process tail alone at 3840x2160 (nothing else on the GPU): run under tools/kstats_cmd.sh for per-kernel times."""
import sys
sys.path.insert(0, ".")
from renderer_amd.api import Renderer, default_settings, load_hip_capi
from tests import helpers as hp
W, H = 3840, 2160
r = Renderer(load_hip_capi(), W, H, default_settings(msaa=1))
hdr = hp.random_hdr(W, H, seed=1)
for _ in range(12):
    r.post_process(hdr, 1.5)
r.close()
