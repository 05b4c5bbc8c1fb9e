"""Shared helpers of the parity tests."""
import numpy as np

from renderer_amd import scenes
from renderer_amd.api import Renderer, default_settings


def psnr(a, b):
    mse = np.mean((a.astype(np.float64) - b.astype(np.float64)) ** 2)
    return 99.0 if mse == 0 else 10.0 * np.log10(255.0 ** 2 / mse)


def make_pair(orc_capi, hip_capi, W, H, **settings):
    s1, s2 = default_settings(**settings), default_settings(**settings)
    return Renderer(orc_capi, W, H, s1, keep_intermediates=True), Renderer(hip_capi, W, H, s2, keep_intermediates=True)


def small_settings(msaa=1):
    return dict(msaa=msaa, directional_shadow_map_resolution=512, point_shadow_map_resolution=128)


def random_hdr(W, H, seed):
    rng = np.random.default_rng(seed)
    base = rng.random((H, W, 4), dtype=np.float32) ** 3 * 4.0
    # a few very bright and a few black pixels
    mask = rng.random((H, W)) > 0.995
    base[mask, :3] *= 40.0
    base[rng.random((H, W)) > 0.99] = 0.0
    return base


def assert_image_close(out_ref, out_hip, max_err=1, min_psnr=45.0):
    """The tolerance of the 8-bit output: <= 1/255 per channel and >= 45 dB PSNR."""
    diff = np.abs(out_ref.astype(np.int32) - out_hip.astype(np.int32))
    assert diff.max() <= max_err, f"max 8-bit error {diff.max()} at {np.unravel_index(diff.argmax(), diff.shape)}"
    p = psnr(out_ref, out_hip)
    assert p >= min_psnr, f"PSNR {p:.2f} dB"
    return diff.max(), p
