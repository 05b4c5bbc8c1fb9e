"""Randomised parity runs of the frame's tail alone (bloom chain, tonemap, luminance): rdr_post_process of the HIP library
against the CPU checker on images of random size (1 x 1 upwards, both paths of the bloom chain) and content (dark,
bright, a few enormous or zero texels, optionally infinities and NaNs).

  python tools/fuzz_post.py [first_seed] [count]

Bloom levels must be equal bit for bit, the 8-bit frame within one step, the luminance image and its average within
1e-5 (or non-finite on both sides)."""
import sys

import numpy as np

sys.path.insert(0, __file__.rsplit("/tools/", 1)[0])
from oracle import orc  # noqa: E402
from renderer_amd.api import Renderer, default_settings, load_hip_capi  # noqa: E402


def image(rng, W, H, special):
    kind = int(rng.integers(0, 4))
    a = rng.random((H, W, 4), dtype=np.float32)
    a = [a * 0.05, a ** 3 * 4.0, a * 60.0, np.exp(rng.normal(0, 3.0, (H, W, 4))).astype(np.float32)][kind]
    n = max(1, W * H // 50)
    ys, xs = rng.integers(0, H, n), rng.integers(0, W, n)
    a[ys, xs, :3] = rng.choice([0.0, 1e-7, 500.0, 65504.0], (n, 1))
    if special:
        ys, xs = rng.integers(0, H, 2), rng.integers(0, W, 2)
        a[ys[0], xs[0], int(rng.integers(0, 3))] = np.inf
        a[ys[1], xs[1], int(rng.integers(0, 3))] = np.nan
    a[..., 3] = 1.0
    with np.errstate(over="ignore"):   # values beyond binary16 become infinities, on purpose, in the special images
        return np.minimum(a, 65504.0).astype(np.float16).astype(np.float32) if not special else a.astype(np.float16).astype(np.float32)


def one(seed, orc_capi, hip_capi, special=False):
    rng = np.random.default_rng(seed)
    W = int(rng.choice([rng.integers(1, 8), rng.integers(8, 140), rng.integers(128, 700)]))
    H = int(rng.choice([rng.integers(1, 8), rng.integers(8, 140), rng.integers(128, 500)]))
    st = dict(hdr_output_enabled=int(rng.random() < 0.3), gamma=float(rng.choice([1.0, 2.2, 2.4])))
    ro = Renderer(orc_capi, W, H, default_settings(**st), keep_intermediates=True)
    rh = Renderer(hip_capi, W, H, default_settings(**st), keep_intermediates=True)
    hdr = image(rng, W, H, special)
    exposure = float(rng.uniform(-3, 3))
    a, la = ro.post_process(hdr, exposure)
    b, lb = rh.post_process(hdr, exposure)
    what = f"seed {seed} {W}x{H} {st} exposure {exposure:.2f}"
    for level in range(6):
        bo, bh = ro.read_bloom(level), rh.read_bloom(level)
        if not np.array_equal(bo, bh, equal_nan=True):
            print(f"{what}: bloom level {level} differs in {int(((bo != bh) & ~(np.isnan(bo) & np.isnan(bh))).sum())} of {bo.size} values")
            return False
    if a.dtype == np.float16:
        a, b = np.nan_to_num(a.astype(np.float32)) * 255.0, np.nan_to_num(b.astype(np.float32)) * 255.0
    d = np.abs(a.astype(np.float64) - b.astype(np.float64))
    if d.max() > 1.0:
        print(f"{what}: output apart by {d.max():.2f} steps in {int((d > 1).sum())} values")
        return False
    lo, lh = ro.read_log_luminance(), rh.read_log_luminance()
    fin = np.isfinite(lo)
    if not np.array_equal(fin, np.isfinite(lh)) or not np.allclose(lo[fin], lh[fin], atol=2e-5, rtol=1e-6):
        print(f"{what}: log-luminance image differs")
        return False
    if np.isfinite(la) != np.isfinite(lb) or (np.isfinite(la) and abs(la - lb) > 1e-4):
        print(f"{what}: average log-luminance {la} / {lb}")
        return False
    ro.close(); rh.close()
    return True


def main():
    first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    count = int(sys.argv[2]) if len(sys.argv) > 2 else 100
    special = len(sys.argv) > 3 and sys.argv[3] == "special"
    orc_capi, hip_capi = orc.capi(), load_hip_capi()
    bad = 0
    for seed in range(first, first + count):
        if not one(seed, orc_capi, hip_capi, special):
            bad += 1
    print(f"{count} images, {bad} with a difference")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
