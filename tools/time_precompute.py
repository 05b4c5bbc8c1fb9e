"""Wall-clock of the environment precompute at BASELINE config 4 size (4096x2048 panorama -> 1024^2 faces, 256^2
irradiance, 512^2 x 10 prefiltered levels, 400^2 table) on the HIP library; each call synchronises before returning."""
import json
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from renderer_amd.api import Renderer, default_settings, load_hip_capi  # noqa: E402

r = Renderer(load_hip_capi(), 64, 64, default_settings(msaa=1))
rng = np.random.default_rng(1)
pano = rng.random((2048, 4096, 4), dtype=np.float32)
out = {}
for rep in range(3):
    t0 = time.perf_counter(); r.set_environment_equirect(pano, 1024); t1 = time.perf_counter()
    r.generate_brdf_lut(400, 1024); t2 = time.perf_counter()
    out = {"equirect_and_bakes_ms": (t1 - t0) * 1e3, "brdf_lut_ms": (t2 - t1) * 1e3, "panorama_upload_bytes": pano.nbytes}
print(json.dumps(out))
