"""Environment precompute: panorama -> cube faces -> irradiance / GGX-prefiltered cubes, and the split-sum BRDF table.

The reference performs the two bakes (Renderer/VulkanRendererEnvironment.cpp:193-250, 324-391) and ships the table as
Renderer/textures/ibl_brdf_lut.png; the panorama conversion and the table generator have no reference code, so the
table is pinned by that PNG and the conversion by analytic panoramas. On the GPU every output is compared with the
CPU restatement bit for bit (decoded B10G11R11 texels, which are equal exactly when the packed words are, and 8-bit table entries)."""
import os

import numpy as np
import pytest
from PIL import Image

from renderer_amd import _binding as B
from renderer_amd.api import Renderer, default_settings

ERR_INVALID = 1  # RDR_ERR_INVALID
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def face_directions(res):
    """Unit directions through the texel centres of the six faces (Vulkan face tables inverted)."""
    t = (2 * np.arange(res) + 1) / res - 1
    sc, tc = np.meshgrid(t, t)
    one = np.ones_like(sc)
    d = [(one, -tc, -sc), (-one, -tc, sc), (sc, one, tc), (sc, -one, -tc), (sc, -tc, one), (-sc, -tc, -one)]
    d = np.stack([np.stack(f, -1) for f in d])
    return d / np.linalg.norm(d, axis=-1, keepdims=True)


def panorama_of(fn, w, h):
    """Panorama whose texel centres hold fn(direction): u = atan2(x, -z) / 2pi + 1/2, v = angle from +Y / pi."""
    u, v = (np.arange(w) + 0.5) / w, (np.arange(h) + 0.5) / h
    phi, theta = (u[None, :] - 0.5) * 2 * np.pi, v[:, None] * np.pi
    d = np.stack([np.sin(theta) * np.sin(phi), np.cos(theta) * np.ones_like(phi), -np.sin(theta) * np.cos(phi)], -1)
    rgb = fn(d)
    return np.concatenate([rgb, np.ones_like(rgb[..., :1])], -1).astype(np.float32)


def smooth_sky(d):
    return np.stack([0.6 + 0.4 * d[..., 0], 0.5 + 0.45 * d[..., 1], 0.7 + 0.25 * d[..., 2] * d[..., 0]], -1)


def sun_sky(d, seed=3):
    rng = np.random.default_rng(seed)
    sun = np.array([0.3, 0.8, -0.52]); sun /= np.linalg.norm(sun)
    c = np.clip((d @ sun), 0, 1) ** 64
    base = smooth_sky(d) * 0.4
    return base + c[..., None] * np.array([40.0, 35.0, 28.0]) + rng.random(d.shape[:-1] + (1,)) * 0.05


def make(capi, **kw):
    return Renderer(capi, 32, 32, default_settings(msaa=1, **kw))


def test_generated_table_against_the_reference_png(orc_capi):
    ref = np.asarray(Image.open(os.path.join(GOLDEN, "ibl_brdf_lut.png")).convert("RGB")).astype(np.int32)
    r = make(orc_capi)
    r.generate_brdf_lut(400, 1024)
    lut = r.read_brdf_lut().astype(np.int32)
    assert lut.shape == (400, 400, 4) and (lut[..., 3] == 255).all()
    d = np.abs(lut[..., :2] - ref[..., :2])
    assert d.mean() < 0.5 and np.percentile(d, 99) <= 6 and np.percentile(d, 90) <= 1, (d.mean(), np.percentile(d, 99), d.max())
    assert np.abs(lut[::-1, :, :2] - ref[..., :2]).mean() > 20, "row 0 must be roughness 1"
    r.close()


def test_table_limits(orc_capi):
    """Known limits of the split sum: scale + bias -> 1 for a smooth surface seen head-on; bias -> 0 there."""
    r = make(orc_capi)
    r.generate_brdf_lut(64, 512)
    lut = r.read_brdf_lut().astype(np.float64) / 255
    smooth_head_on = lut[-1, -1]
    assert abs(smooth_head_on[0] + smooth_head_on[1] - 1.0) < 0.02 and smooth_head_on[1] < 0.01
    assert lut[-1, 0, 1] > 0.5, "grazing view of a smooth surface is mostly Fresnel bias"
    assert (np.diff(lut[:, -1, 0]) >= -1 / 255).all(), "head-on scale grows as roughness falls"
    r.close()


def test_argument_checks(orc_capi):
    r = make(orc_capi)
    assert orc_capi.generate_brdf_lut(r._ctx, 0, 16) == ERR_INVALID
    assert orc_capi.generate_brdf_lut(r._ctx, 16, 0) == ERR_INVALID
    assert orc_capi.read_brdf_lut(r._ctx, None, None, None) == ERR_INVALID
    pano = np.zeros((4, 8, 4), np.float32)
    assert orc_capi.set_environment_equirect(r._ctx, B.fptr(pano), 8, 4, 0) == ERR_INVALID
    assert orc_capi.set_environment_equirect(r._ctx, None, 8, 4, 16) == ERR_INVALID
    r.close()


def test_constant_panorama_gives_constant_cubes(orc_capi):
    r = make(orc_capi)
    pano = np.empty((16, 32, 4), np.float32)
    pano[...] = (0.5, 0.25, 2.0, 1.0)
    r.set_environment_equirect(pano, 32)
    env = r.read_env_cube(0)
    assert np.allclose(env, (0.5, 0.25, 2.0), rtol=0, atol=0)
    pre = r.read_env_cube(2, 3)
    assert np.allclose(pre, (0.5, 0.25, 2.0), rtol=2.0 ** -5)
    r.close()


def test_panorama_orientation_and_interpolation(orc_capi):
    """A smooth function of direction survives the resampling: every cube texel equals the function at its own
    direction to within the bilinear error plus packed-float truncation, so axes and handedness are right."""
    r = make(orc_capi)
    r.set_environment_equirect(panorama_of(smooth_sky, 512, 256), 64)
    env = r.read_env_cube(0).astype(np.float64)
    want = smooth_sky(face_directions(64))
    assert np.abs(env - want).max() < 0.02
    # packed storage truncates, so the stored value never exceeds the interpolated one by more than rounding
    assert (env <= want + 2e-3).all()
    r.close()


def test_pole_and_seam_texels(orc_capi):
    """Panorama column 0 and column w-1 are neighbours (u wraps); rows clamp at the poles."""
    r = make(orc_capi)
    w, h = 64, 32
    pano = np.zeros((h, w, 4), np.float32)
    pano[:, 0, 0] = 1.0          # red stripe right of the seam
    pano[:, w - 1, 1] = 1.0      # green stripe left of it
    pano[0, :, 2] = 1.0          # blue cap at +Y
    r.set_environment_equirect(pano, 64)
    env = r.read_env_cube(0).astype(np.float64)
    d = face_directions(64)
    u = np.arctan2(d[..., 0], -d[..., 2]) / (2 * np.pi) + 0.5
    v = np.arccos(np.clip(d[..., 1], -1, 1)) / np.pi
    mid = (v > 0.2) & (v < 0.8)
    seam = mid & ((u < 0.2 / w) | (u > 1 - 0.2 / w))
    assert seam.sum() > 0
    assert (np.abs(env[seam][:, 0] + env[seam][:, 1] - 1.0) < 0.02).all(), "seam texels blend the two stripes"
    far = mid & (u > 2.0 / w) & (u < 1 - 2.0 / w)
    assert (env[far][:, :2] == 0).all()
    top = v < 0.25 / h
    assert top.sum() > 0 and (env[top][:, 2] > 0.98).all()
    r.close()


@pytest.mark.gpu
def test_hip_table_matches_the_restatement(orc_capi, hip_capi):
    for size, samples in ((400, 1024), (33, 100)):
        ro, rh = make(orc_capi), make(hip_capi)
        ro.generate_brdf_lut(size, samples); rh.generate_brdf_lut(size, samples)
        assert np.array_equal(ro.read_brdf_lut(), rh.read_brdf_lut()), (size, samples)
        ro.close(); rh.close()


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(1024, 512, 256), (300, 157, 65)])
def test_hip_environment_precompute_matches_the_restatement(orc_capi, hip_capi, shape):
    """BASELINE config 4 at reduced panorama size, and a ragged one (odd sizes, non-power-of-two faces)."""
    w, h, res = shape
    pano = panorama_of(sun_sky, w, h)
    ro, rh = make(orc_capi), make(hip_capi)
    ro.set_environment_equirect(pano, res); rh.set_environment_equirect(pano, res)
    assert np.array_equal(ro.read_env_cube(0), rh.read_env_cube(0)), "cube faces"
    assert np.array_equal(ro.read_env_cube(1), rh.read_env_cube(1)), "irradiance"
    for level in range(10):
        assert np.array_equal(ro.read_env_cube(2, level), rh.read_env_cube(2, level)), ("prefiltered", level)
    ro.close(); rh.close()


@pytest.mark.gpu
def test_hip_full_size_precompute_properties(hip_capi):
    """BASELINE config 4 at full size (4096x2048 panorama, 1024^2 faces): checked through properties since the CPU
    restatement of the 512^2 x 10-level prefilter at this size is the slow part, not the conversion."""
    rh = make(hip_capi)
    pano = panorama_of(smooth_sky, 4096, 2048)
    rh.set_environment_equirect(pano, 1024)
    env = rh.read_env_cube(0).astype(np.float64)
    want = smooth_sky(face_directions(1024))
    assert np.abs(env - want).max() < 0.02
    irr = rh.read_env_cube(1)
    assert np.isfinite(irr).all() and irr.min() >= 0
    p0, p9 = rh.read_env_cube(2, 0).reshape(-1, 3), rh.read_env_cube(2, 9).reshape(-1, 3)
    assert p0.shape[0] == 6 * 512 * 512 and p9.shape[0] == 6
    assert p9.std(axis=0).max() < p0.std(axis=0).max(), "rougher levels are smoother"
    rh.close()
