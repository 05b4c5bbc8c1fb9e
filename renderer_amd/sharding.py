"""Frame-batch sharding across the GPUs of one node.

Frames (and camera views) are independent units of the per-frame pipeline: the
scene is replicated on every GPU, frame k of a job goes to rank k mod R, and
the only exchange is the gather of the finished RGBA8 images to rank 0. One
process per GPU, `torch.distributed` with the "nccl" backend (RCCL over xGMI);
the same code runs with "gloo" and host tensors, which is how it is tested
without GPUs.

Up to three frames are in flight on a GPU: frame k is only enqueued, the host
goes on to build and enqueue the frames behind it, and frame k's image and
luminance are collected when the renderer reports it complete (two calls later
at the latest, or after wait_frame). The gather of frame k is issued
asynchronously at that point and waited for only when its staging slot is
needed again, so it overlaps the rendering of the next frames.
Rank 0 receives over up to R-1 distinct xGMI links at once (a 4K RGBA8 frame is
33 MB, about 0.2 ms per link), which is why a plain gather is used rather than
a ring all-gather.

The reference renders on one device only (one vk::Queue,
Renderer/VulkanRenderer.cpp:548); this module has no counterpart there.
"""
import numpy as np


def frames_of_rank(n_frames, rank, world):
    """Frame indices rendered by `rank`: k with k mod world == rank, in order."""
    if not (0 <= rank < world):
        raise ValueError(f"rank {rank} outside world of {world}")
    return list(range(rank, int(n_frames), world))


def rounds(n_frames, world):
    """Number of gather rounds: every rank takes part in each of them (ranks without a frame send padding)."""
    return (int(n_frames) + world - 1) // world


class FrameSharder:
    """Renders this rank's share of a frame batch and gathers the images to rank 0.

    renderer  : api.Renderer bound to this rank's GPU
    dist      : the initialised torch.distributed module, or None for a single process
    device    : torch device of the staging tensors ("cuda:N" with nccl, "cpu" with gloo)
    depth     : staging slots: a slot is busy from the call that renders into it until the gather of its frame has
                finished; with fewer than three the renderer is made to wait for a frame before its slot is reused
    pipelined : keep frames in flight on the renderer (a frame is only enqueued; its results are collected once the
                renderer reports it complete). Default: on for device staging tensors, off for host ones.
    After `run`, `gather_wait_s` is the time this rank spent waiting for gathers it had issued (rank 0: for the frames
    of all ranks to arrive) and `frames_rendered` the frames it rendered itself.
    """

    def __init__(self, renderer, dist=None, device="cpu", depth=2, pipelined=None):
        import torch
        self.torch = torch
        self.r = renderer
        self.dist = dist
        self.rank = dist.get_rank() if dist else 0
        self.world = dist.get_world_size() if dist else 1
        self.device = torch.device(device)
        self.pipelined = (self.device.type == "cuda") if pipelined is None else bool(pipelined)
        self.gather_wait_s = 0.0
        self.frames_rendered = 0
        self.depth = max(1, int(depth))
        H, W = renderer.height, renderer.width
        # the renderer's output image: RGBA8, or RGBA16F for a context created with hdr_output_enabled
        pixel = torch.float16 if renderer.settings.hdr_output_enabled else torch.uint8
        self.slots = [torch.empty((H, W, 4), dtype=pixel, device=self.device) for _ in range(self.depth)]
        self.recv = None
        if dist and self.rank == 0:
            self.recv = [[torch.empty((H, W, 4), dtype=pixel, device=self.device) for _ in range(self.world)]
                         for _ in range(self.depth)]
        self.pending = [None] * self.depth   # (work handle, round index) per slot
        self.luminance = {}

    # -- one frame ---------------------------------------------------------
    def _render_into(self, tensor, frame_params):
        """Renders into `tensor`. A device tensor only gets the frame enqueued (returns None: the frame's results are
        picked up by `_collect` once the renderer reports it complete); a host
        tensor is rendered to completion and its average log-luminance returned."""
        self.frames_rendered += 1
        if self.pipelined:
            self.r.render_frame(device_ptr=tensor.data_ptr(), **frame_params)
            return None
        out = tensor.numpy()
        _, avg = self.r.render_frame(out=out, **frame_params)
        return avg

    def _collect(self, k, first_index, after_render):
        """Frame k of this rank has finished on its GPU: record its luminance, run the per-frame hook."""
        s = self.r.frame_stats()
        if s.frame_index != first_index + k // self.world:
            raise RuntimeError(f"frame statistics out of step: expected frame {first_index + k // self.world} of the context, got {s.frame_index}")
        self.luminance[k] = s.avg_log_luminance
        if after_render is not None:
            after_render(k)

    def _retire(self, slot, on_frames):
        if self.pending[slot] is None:
            return
        work, rnd, n_valid = self.pending[slot]
        if work is not None:
            import time
            t0 = time.perf_counter()
            work.wait()
            if self.device.type == "cuda":
                # wait() only orders torch's current stream behind the gather; the renderer writes this slot from its
                # own stream, so the host makes sure the gather (issued a frame ago) has really finished
                self.torch.cuda.current_stream(self.device).synchronize()
            self.gather_wait_s += time.perf_counter() - t0
        self.pending[slot] = None
        if on_frames is not None and self.rank == 0:
            imgs = self.recv[slot] if self.recv is not None else [self.slots[slot]]
            for src in range(n_valid):
                on_frames(rnd * self.world + src, imgs[src])

    def run(self, n_frames, set_frame, frame_params=None, on_frame=None, after_render=None):
        """Renders frames 0..n_frames-1 of the job. `set_frame(renderer, k)` positions the camera / nodes for
        frame k. On rank 0 `on_frame(k, image_tensor)` is called once per frame, in increasing k, with the
        gathered image (valid only during the call). `after_render(k)` runs on the rendering rank right after
        frame k has finished on its GPU (per-frame statistics). Returns {frame index: avg log-luminance} of
        this rank."""
        frame_params = dict(frame_params or {})
        n_rounds = rounds(n_frames, self.world)
        first_index = self.r.frame_stats().frame_index + 1 if self.pipelined else 0  # index of this job's frame 0
        in_flight = []  # (round, frame or None) enqueued on the GPU, oldest first, whose results have not been collected

        def finish(entry):
            # the frame is complete: collect its results, then hand its image to the gather
            rnd_f, k_f = entry
            if k_f is not None:
                self._collect(k_f, first_index, after_render)
            slot_f = rnd_f % self.depth
            n_valid = min(self.world, n_frames - rnd_f * self.world)
            work = None
            if self.dist:
                work = self.dist.gather(self.slots[slot_f], self.recv[slot_f] if self.rank == 0 else None, dst=0, async_op=True)
            self.pending[slot_f] = (work, rnd_f, n_valid)

        def finish_through(last_round):
            # waits, oldest first, for the frames of the rounds up to `last_round` and finishes them
            while in_flight and in_flight[0][0] <= last_round:
                entry = in_flight.pop(0)
                if entry[1] is not None:
                    self.r.wait_frame(first_index + entry[1] // self.world)
                finish(entry)

        for rnd in range(n_rounds):
            slot = rnd % self.depth
            if self.pipelined:
                finish_through(rnd - self.depth)  # the frame that was rendered into this slot, if it is still in flight
            self._retire(slot, on_frame)
            k = rnd * self.world + self.rank
            avg = None
            if k < n_frames:
                set_frame(self.r, k)
                avg = self._render_into(self.slots[slot], frame_params)
            if self.pipelined:
                # frames in flight: frame k was only enqueued; the call has picked up the results of the frames that
                # have finished by now (the renderer says which: at most one more per call), and those are finished
                # here. A rank without a frame in this round enqueued nothing and waits for what it has in flight.
                in_flight.append((rnd, k if k < n_frames else None))
                if k >= n_frames:
                    finish_through(rnd)
                else:
                    done = self.r.frame_stats().frame_index
                    while in_flight and in_flight[0][1] is not None and first_index + in_flight[0][1] // self.world <= done:
                        finish(in_flight.pop(0))
            else:
                if k < n_frames:
                    self.luminance[k] = avg
                    if after_render is not None:
                        after_render(k)
                n_valid = min(self.world, n_frames - rnd * self.world)
                work = None
                if self.dist:
                    work = self.dist.gather(self.slots[slot], self.recv[slot] if self.rank == 0 else None, dst=0, async_op=True)
                self.pending[slot] = (work, rnd, n_valid)
        finish_through(n_rounds)
        # drain in round order
        order = sorted((p[1], s) for s, p in enumerate(self.pending) if p is not None)
        for _, s in order:
            self._retire(s, on_frame)
        return dict(self.luminance)

    def gather_luminance(self, n_frames):
        """All ranks get the avg log-luminance of every frame of the job (one float per frame), the exchange the
        exposure recurrence needs when frames of one sequence are rendered on different GPUs."""
        torch = self.torch
        n_rounds = rounds(n_frames, self.world)
        mine = torch.zeros(n_rounds, dtype=torch.float32, device=self.device)
        for k, v in self.luminance.items():
            mine[k // self.world] = float(v)
        if not self.dist:
            return mine.cpu().numpy()[:n_frames].copy()
        parts = [torch.empty_like(mine) for _ in range(self.world)]
        self.dist.all_gather(parts, mine)
        table = torch.stack(parts, 1).reshape(-1).cpu().numpy()  # [round][rank] -> frame index
        return table[:n_frames].copy()


def exposure_sequence(avg_log_luminance, delta_time, exposure_bias=0.0, start=0.0):
    """The reference's exposure recurrence over a frame sequence (Renderer/VulkanRenderer.cpp:1075-1082), as a
    function of the per-frame average log-luminance. Frame k uses the luminance of frame k-1 (0 for the first).
    Returns (temporal_luminance[k], exposure[k]) as float32 arrays."""
    n = len(avg_log_luminance)
    tl = np.zeros(n, np.float32)
    ex = np.zeros(n, np.float32)
    cur = np.float32(start)
    prev = np.float32(0.0)
    for k in range(n):
        avg = min(prev, np.float32(10.0))
        cur = np.float32(float(cur) + (float(np.exp2(np.float32(avg))) - float(cur)) * (1.0 - np.exp(-float(delta_time))))
        tl[k] = cur
        e = np.float32(1.03) - np.float32(2.0) / (np.log2(np.float32(cur + np.float32(1.0))) + np.float32(2.0))
        ex[k] = np.float32(0.18) / np.float32(e) + np.float32(exposure_bias)
        prev = np.float32(avg_log_luminance[k])
    return tl, ex
