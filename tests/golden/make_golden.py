"""Regenerates the golden vectors under tests/golden/ from the CPU restatement (oracle/).

    python tests/golden/make_golden.py

The reference ships no golden images and cannot be built or run here, so these vectors do not come from the
reference itself: they freeze the outputs of the restatement (which tests/test_oracle_kat.py pins to the reference's
source) so that a change in either the restatement or the HIP library is caught at once. The only file here that
does come from the reference is ibl_brdf_lut.png, its shipped split-sum BRDF table (a data file, copied unchanged
from Renderer/textures/).
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

CASES = {
    # name: (W, H, msaa, directional shadow res, point shadow res)
    "small_msaa1": (96, 54, 1, 128, 32),
    "small_msaa4": (96, 54, 4, 128, 32),
}
FRAME = dict(exposure_mode=1, temporal_luminance=0.5)


def build(r):
    from renderer_amd import scenes
    scenes.small_scene(r, tex_size=32, env_res=16)


def settings_of(case):
    from renderer_amd.api import default_settings
    W, H, msaa, dres, pres = CASES[case]
    return W, H, default_settings(msaa=msaa, directional_shadow_map_resolution=dres, point_shadow_map_resolution=pres)


def render_case(capi, case):
    """All compared outputs of one case from the library behind `capi`."""
    from renderer_amd.api import Renderer
    W, H, st = settings_of(case)
    r = Renderer(capi, W, H, st)
    build(r)
    out, avg = r.render_frame(**FRAME)
    res = {
        "rgba8": out, "avg_log_luminance": np.float32(avg), "depth": r.read_depth(), "visibility": r.read_visibility(),
        "cascade0": r.read_directional_shadow(0), "cascade3": r.read_directional_shadow(3),
        "point_0_2": r.read_point_shadow(0, 2), "hdr": r.read_hdr(), "bloom3": r.read_bloom(3),
        "prefiltered4": r.read_env_cube(2, 4), "irradiance_face2": r.read_env_cube(1, 0)[2, ::8, ::8],
    }
    vp, d = r.read_cascades()
    res["cascade_viewproj"], res["cascade_depth"] = vp, d
    r.close()
    return res


def post_case(capi, seed=7, W=80, H=48, exposure=1.5):
    from renderer_amd.api import Renderer, default_settings
    rng = np.random.default_rng(seed)
    hdr = (rng.random((H, W, 4), dtype=np.float32) ** 3 * 5.0).astype(np.float32)
    hdr[rng.random((H, W)) > 0.99, :3] *= 30.0
    r = Renderer(capi, W, H, default_settings(), keep_intermediates=True)
    out, avg = r.post_process(hdr, exposure)
    res = {"hdr_in": hdr, "rgba8": out, "avg_log_luminance": np.float32(avg), "bloom0": r.read_bloom(0), "bloom5": r.read_bloom(5)}
    r.close()
    return res


def main():
    from oracle import orc
    capi = orc.capi()
    for case in CASES:
        np.savez_compressed(os.path.join(HERE, case + ".npz"), **render_case(capi, case))
    np.savez_compressed(os.path.join(HERE, "post_80x48.npz"), **post_case(capi))
    for f in sorted(os.listdir(HERE)):
        print(f, os.path.getsize(os.path.join(HERE, f)))


if __name__ == "__main__":
    main()
