"""Frames in flight against frames one at a time, on the HIP library alone: the same random scene and the same changes
between frames (tools/fuzz_parity.py) rendered twice - into host memory, every frame complete before the next call,
and into a rotation of device buffers with nothing waited for in between (rdr_render_frame with a device destination
returns as soon as the frame is enqueued, two or three frames in flight). Every frame of the second run must equal
the first run's, byte for byte: a buffer of a frame slot reused too early, an upload that overtakes the frame still
reading the old contents, a cached shadow view that should have been redrawn, would all show here.

  python tools/fuzz_pipeline.py [first_seed] [count] [frames]        (RDR_FRAMES_IN_FLIGHT=2 or 3 in the environment)"""
import sys

import numpy as np
import torch

sys.path.insert(0, __file__.rsplit("/tools/", 1)[0])
from tools import fuzz_parity as F  # noqa: E402
from renderer_amd.api import Renderer, default_settings, load_hip_capi  # noqa: E402


def one(seed, hip_capi, frames):
    rng = np.random.default_rng(seed + 10 ** 6)
    W, H = int(rng.integers(17, 400)), int(rng.integers(17, 300))
    msaa = int(rng.choice([1, 4]))
    st = dict(msaa=msaa, directional_shadow_map_resolution=int(rng.choice([64, 200, 512])), point_shadow_map_resolution=int(rng.choice([16, 50, 128])),
              point_shadow_map_count=int(rng.integers(0, 4)), dynamic_shadows_enabled=int(rng.random() < 0.8))
    ra = Renderer(hip_capi, W, H, default_settings(**st))
    rb = Renderer(hip_capi, W, H, default_settings(**st))
    ma, mb = F.build(ra, seed), F.build(rb, seed)
    want = []
    for frame in range(frames):
        if frame >= 1:
            F.change(ra, ma, seed, frame)
        out, _ = ra.render_frame(exposure_mode=1, temporal_luminance=0.5)
        want.append(out.copy())
    bufs = [torch.zeros((H, W, 4), dtype=torch.uint8, device="cuda") for _ in range(frames)]
    for frame in range(frames):
        if frame >= 1:
            F.change(rb, mb, seed, frame)
        rb.render_frame(exposure_mode=1, temporal_luminance=0.5, device_ptr=bufs[frame].data_ptr())
    rb.synchronize()
    ok = True
    for frame in range(frames):
        got = bufs[frame].cpu().numpy()
        if not np.array_equal(got, want[frame]):
            d = np.abs(got.astype(int) - want[frame].astype(int)).max(-1)
            print(f"seed {seed} frame {frame}: {int((d > 0).sum())} of {d.size} pixels differ, largest step {int(d.max())} ({W}x{H} msaa {msaa} {st})")
            ok = False
    ra.close(); rb.close()
    return ok


def main():
    first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    count = int(sys.argv[2]) if len(sys.argv) > 2 else 50
    frames = int(sys.argv[3]) if len(sys.argv) > 3 else 6
    hip_capi = load_hip_capi()
    bad = 0
    for seed in range(first, first + count):
        bad += 0 if one(seed, hip_capi, frames) else 1
    print(f"{count} scenes of {frames} frames, {bad} with a difference")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
