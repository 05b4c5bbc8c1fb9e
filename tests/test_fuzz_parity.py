"""Random scenes (tools/fuzz_parity.py): the HIP library against the CPU checker, one context after the other in one
process. Depth, visibility and every shadow map bit for bit; the 8-bit frame within one step but for single pixels on
a discontinuity. A frame shorter than 32 pixels has a bloom level no down-sample writes (the reference's dispatch
extent is W >> l x H >> l, VulkanRendererBloom.cpp:82): it must read as zero whatever memory the context was given,
which a run of contexts in one process is what exposes."""
import pytest

from tools import fuzz_parity as F

pytestmark = pytest.mark.gpu


def test_random_scenes(orc_capi, hip_capi):
    worst = 0.0
    for seed in range(0, 16):
        ok, share = F.one(seed, orc_capi, hip_capi)
        assert ok, seed
        assert share <= 0.002, (seed, share)
        worst = max(worst, share)
    print("largest share of pixels further apart than one step:", worst)


def test_random_images_through_the_tail(orc_capi, hip_capi):
    """tools/fuzz_post.py: bloom chain, tonemap and luminance on images from 1 x 1 upwards, dark to 65504, then with
    infinities and NaNs in them: bloom levels bit for bit, the output within one step."""
    from tools import fuzz_post as P
    for seed in range(0, 80):
        assert P.one(seed, orc_capi, hip_capi), seed
    for seed in range(1000, 1040):
        assert P.one(seed, orc_capi, hip_capi, special=True), seed


def test_random_environments(orc_capi, hip_capi):
    """tools/fuzz_bake.py: faces from 1 x 1 upwards or a panorama, dim skies to 60000-nit suns, through the bakes: every
    cube and every generated BRDF table bit for bit."""
    from tools import fuzz_bake as K
    for seed in range(0, 8):
        assert K.one(seed, orc_capi, hip_capi), seed


def test_frames_in_flight_equal_frames_one_at_a_time(hip_capi):
    """tools/fuzz_pipeline.py: random scenes with changes between frames, rendered into host memory frame by frame and
    into device buffers with nothing waited for in between: the same bytes."""
    from tools import fuzz_pipeline as P
    for seed in range(0, 20):
        assert P.one(seed, hip_capi, 6), seed
