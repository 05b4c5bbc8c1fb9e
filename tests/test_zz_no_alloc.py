"""The entry point that allocates nothing (rdr_post_process_noalloc) against rdr_post_process, through the C ABI:
same bytes, same average luminance, on two calls in a row, the second on a smaller image (the scratch array inside
the library still holds the first call's contents), then a third on a ragged size; sizes beyond the stated maximum are
refused."""
import ctypes as C

import numpy as np
import pytest

from renderer_amd import _binding as B
from renderer_amd.api import Renderer, default_settings
from tests import helpers as hp

pytestmark = pytest.mark.gpu


def test_no_alloc_post_process_matches_post_process(hip_capi):
    import torch
    dev = torch.device("cuda:0")
    ctx = Renderer(hip_capi, 64, 64, default_settings())  # the sizes passed to the call need not be the context's
    # (the last one is shorter than 32 pixels: its smallest bloom level is written by no down-sample and must read as zero,
    # not as what the call before left in the array)
    for (W, H), exposure in (((1920, 1080), 1.25), ((640, 360), 0.5), ((333, 277), -0.75), ((138, 26), 0.5)):
        hdr = hp.random_hdr(W, H, seed=W + H).astype(np.float16)
        ref = Renderer(hip_capi, W, H, default_settings())
        want, want_avg = ref.post_process(hdr.astype(np.float32), exposure)
        ref.close()
        d_hdr = torch.from_numpy(hdr.view(np.int16)).to(dev)
        d_out = torch.zeros((H, W, 4), dtype=torch.uint8, device=dev)
        d_avg = torch.zeros(1, dtype=torch.float32, device=dev)
        torch.cuda.synchronize()
        rc = hip_capi.post_process_noalloc(ctx._ctx, d_hdr.data_ptr(), W, H, exposure, d_out.data_ptr(), d_avg.data_ptr(), None)
        assert rc == B.RDR_OK, hip_capi.last_error(ctx._ctx)
        ctx.synchronize()
        assert np.array_equal(d_out.cpu().numpy(), want), f"{W}x{H}"
        got_avg = float(d_avg.cpu()[0])
        assert got_avg == want_avg or (np.isnan(got_avg) and np.isnan(want_avg)), f"{W}x{H}: {got_avg} vs {want_avg}"
    # beyond the maximum: refused with a message, nothing launched
    d_big = torch.zeros((1200, 1920, 4), dtype=torch.int16, device=dev)
    d_out = torch.zeros((1200, 1920, 4), dtype=torch.uint8, device=dev)
    rc = hip_capi.post_process_noalloc(ctx._ctx, d_big.data_ptr(), 1920, 1200, 0.0, d_out.data_ptr(), d_avg.data_ptr(), None)
    assert rc == B.RDR_ERR_CAPACITY and b"RDR_NOALLOC_MAX_PIXELS" in hip_capi.last_error(ctx._ctx)
    assert hip_capi.post_process_noalloc(ctx._ctx, d_big.data_ptr(), 1, B.NOALLOC_MAX_PIXELS, 0.0, d_out.data_ptr(), d_avg.data_ptr(), None) == B.RDR_ERR_CAPACITY
    assert hip_capi.post_process_noalloc(ctx._ctx, None, 16, 16, 0.0, d_out.data_ptr(), d_avg.data_ptr(), None) == B.RDR_ERR_INVALID
    ctx.close()
    # a context whose output image is RGBA16F (hdr_output_enabled): 8 bytes per pixel, the same values as its rdr_post_process
    W, H, exposure = 321, 200, 0.25
    ctx = Renderer(hip_capi, W, H, default_settings(hdr_output_enabled=1))
    hdr = hp.random_hdr(W, H, seed=5).astype(np.float16)
    want, want_avg = ctx.post_process(hdr.astype(np.float32), exposure)
    d_hdr = torch.from_numpy(hdr.view(np.int16)).to(dev)
    d_out = torch.zeros((H, W, 4), dtype=torch.float16, device=dev)
    torch.cuda.synchronize()
    rc = hip_capi.post_process_noalloc(ctx._ctx, d_hdr.data_ptr(), W, H, exposure, d_out.data_ptr(), d_avg.data_ptr(), None)
    assert rc == B.RDR_OK, hip_capi.last_error(ctx._ctx)
    ctx.synchronize()
    assert np.array_equal(d_out.cpu().numpy().view(np.uint16), want.view(np.uint16)) and float(d_avg.cpu()[0]) == want_avg
    assert hip_capi.post_process_noalloc(ctx._ctx, d_hdr.data_ptr(), W, H, exposure, d_out.data_ptr() + 8, d_avg.data_ptr(), None) == B.RDR_ERR_INVALID
    ctx.close()
