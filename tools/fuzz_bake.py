"""Randomised parity runs of the environment precompute: six faces of random size (1 x 1 upwards) and content, or a
panorama of random size, through setEnvironmentMap's bakes (VulkanRendererEnvironment.cpp:193-250, 324-391) on the HIP
library and the CPU checker; generated BRDF tables of random size. Every cube (environment, irradiance, the ten
prefiltered levels) and the table must be equal bit for bit.

  python tools/fuzz_bake.py [first_seed] [count]"""
import sys

import numpy as np

sys.path.insert(0, __file__.rsplit("/tools/", 1)[0])
from oracle import orc  # noqa: E402
from renderer_amd.api import Renderer, default_settings, load_hip_capi  # noqa: E402


def content(rng, shape):
    kind = int(rng.integers(0, 4))
    a = rng.random(shape, dtype=np.float32)
    a = [a * 0.01, a, a ** 4 * 50.0, np.exp(rng.normal(0, 2.5, shape)).astype(np.float32)][kind]
    flat = a.reshape(-1, a.shape[-1])
    n = max(1, flat.shape[0] // 100)
    flat[rng.integers(0, flat.shape[0], n), :3] = rng.choice([0.0, 2000.0, 60000.0], (n, 1))   # suns and holes
    a[..., 3] = 1.0
    return a


def one(seed, orc_capi, hip_capi):
    rng = np.random.default_rng(seed)
    ro = Renderer(orc_capi, 16, 16, default_settings(msaa=1))
    rh = Renderer(hip_capi, 16, 16, default_settings(msaa=1))
    what = f"seed {seed}"
    if rng.random() < 0.6:
        res = int(rng.choice([rng.integers(1, 6), rng.integers(6, 48), 2 ** rng.integers(0, 7)]))
        faces = content(rng, (6, res, res, 4))
        what += f" faces {res}"
        for r in (ro, rh):
            r.set_environment_map(faces)
    else:
        w, h, res = int(rng.integers(2, 200)), int(rng.integers(1, 100)), int(rng.integers(1, 40))
        pano = content(rng, (h, w, 4))
        what += f" panorama {w}x{h} -> faces {res}"
        for r in (ro, rh):
            r.set_environment_equirect(pano, res)
    cubes = [(0, 0), (1, 0)] + [(2, l) for l in range(10)]
    for which, level in cubes:
        x, y = ro.read_env_cube(which, level), rh.read_env_cube(which, level)
        if not np.array_equal(x, y, equal_nan=True):
            print(f"{what}: cube {which} level {level} differs in {int(((x != y) & ~(np.isnan(x) & np.isnan(y))).sum())} of {x.size} values")
            return False
    if rng.random() < 0.3:
        size, samples = int(rng.integers(1, 70)), int(rng.choice([1, 7, 64, 200]))
        for r in (ro, rh):
            r.generate_brdf_lut(size, samples)
        if not np.array_equal(ro.read_brdf_lut(), rh.read_brdf_lut()):
            print(f"{what}: BRDF table {size} / {samples} samples differs")
            return False
    ro.close(); rh.close()
    return True


def main():
    first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    count = int(sys.argv[2]) if len(sys.argv) > 2 else 40
    orc_capi, hip_capi = orc.capi(), load_hip_capi()
    bad = 0
    for seed in range(first, first + count):
        ok = one(seed, orc_capi, hip_capi)
        bad += 0 if ok else 1
        print(f"seed {seed}: {'equal' if ok else 'DIFFERS'}", flush=True)
    print(f"{count} environments, {bad} with a difference")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
