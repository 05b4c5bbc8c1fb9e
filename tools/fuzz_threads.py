"""Contexts are independent: several host threads, each with a context of its own on the same GPU, render different
random scenes (tools/fuzz_parity.py) at the same time; every frame must equal the one the same scene gives when it is
rendered alone.
  python tools/fuzz_threads.py [first_seed] [threads] [rounds]"""
import sys
import threading

import numpy as np

sys.path.insert(0, __file__.rsplit("/tools/", 1)[0])
from tools import fuzz_parity as F  # noqa: E402
from renderer_amd.api import Renderer, default_settings, load_hip_capi  # noqa: E402


def frames_of(seed, hip_capi, n_frames, out):
    rng = np.random.default_rng(seed + 10 ** 6)
    W, H = int(rng.integers(17, 400)), int(rng.integers(17, 300))
    st = dict(msaa=int(rng.choice([1, 4])), directional_shadow_map_resolution=int(rng.choice([64, 200, 512])),
              point_shadow_map_resolution=int(rng.choice([16, 50, 128])), point_shadow_map_count=int(rng.integers(0, 4)))
    r = Renderer(hip_capi, W, H, default_settings(**st))
    movable = F.build(r, seed)
    for frame in range(n_frames):
        if frame >= 1:
            F.change(r, movable, seed, frame)
        img, lum = r.render_frame(exposure_mode=1, temporal_luminance=0.5)
        out.append((img.copy(), lum))
    r.close()


def main():
    first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    n_threads = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 10
    hip_capi = load_hip_capi()
    bad = 0
    for rnd in range(rounds):
        seeds = [first + rnd * n_threads + k for k in range(n_threads)]
        alone = {s: [] for s in seeds}
        for s in seeds:
            frames_of(s, hip_capi, 5, alone[s])
        together = {s: [] for s in seeds}
        threads = [threading.Thread(target=frames_of, args=(s, hip_capi, 5, together[s])) for s in seeds]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        for s in seeds:
            same = len(alone[s]) == len(together[s]) and all(np.array_equal(a[0], b[0]) and (a[1] == b[1] or (np.isnan(a[1]) and np.isnan(b[1])))
                                                             for a, b in zip(alone[s], together[s]))
            if not same:
                bad += 1
                print(f"seed {s}: frames rendered beside {n_threads - 1} other contexts differ from the frames rendered alone")
    print(f"{rounds} rounds of {n_threads} threads, {bad} scenes with a difference")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
