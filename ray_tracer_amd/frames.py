"""Frame sequences on one or several GPUs (one process per GPU, `torch.distributed` for the exchange).

A frame is produced in three stages: this rank renders its tiles into a tile-major shard
(`rt_render_shard_device`), the equal-size shards of all ranks are all-gathered, and `rt_untile_device`
writes the row-major frame.  The stages of consecutive frames overlap: rendering runs on one HIP stream,
exchange + untile on another, with two shard buffers and two frame buffers, so that while frame k is
exchanged frame k+1 is already being rendered.  Nothing here computes pixels; torch supplies device
memory, streams, events and the collective.

The reference renders one frame per process run (main.cpp:1212-1288); a sequence of frames is its
`frames/` + `*.mp4` workflow (.gitignore:3-4).
"""
import torch

from . import TILE


class FramePipeline:
    """ctx: a `Context` with the scene uploaded and built.  world/rank: this process's place among the
    GPUs that share every frame (`group`: the process group to gather in, default group when None).
    force_collective runs the all-gather even for a single rank (tests of the stream logic on one GPU)."""

    def __init__(self, ctx, width, height, rank=0, world=1, device=0, float_frame=False, group=None,
                 force_collective=False):
        self.ctx, self.width, self.height, self.rank, self.world = ctx, width, height, rank, world
        self.group = group
        self.dev = torch.device("cuda", device)
        self.gather = world > 1 or force_collective
        p = ctx.params(width, height, rank=rank, world_size=world)
        per_rank, self.total_tiles = ctx.shard_tiles(p)
        n = per_rank * TILE * TILE * 3
        self.render_stream = torch.cuda.Stream(device=self.dev)
        self.exchange_stream = torch.cuda.Stream(device=self.dev)
        z = lambda count, dt: torch.zeros(count, dtype=dt, device=self.dev)
        self.shard8 = [z(n, torch.uint8) for _ in range(2)]
        self.shard = [z(n, torch.float32) for _ in range(2)] if float_frame else [None, None]
        # one gather buffer: exchange and untile of consecutive frames are ordered on the exchange stream
        self.gathered8 = z(world * n, torch.uint8) if self.gather else None
        self.gathered = z(world * n, torch.float32) if (self.gather and float_frame) else None
        self.frame8 = [z(height * width * 3, torch.uint8) for _ in range(2)]
        self.frame = [z(height * width * 3, torch.float32) for _ in range(2)] if float_frame else [None, None]
        self.rendered = [torch.cuda.Event() for _ in range(2)]  # shard b is complete
        self.released = [torch.cuda.Event() for _ in range(2)]  # shard b may be overwritten
        self.done = [torch.cuda.Event() for _ in range(2)]      # frame b is complete
        self.count = 0
        # A stream gets its hardware queue at first use, and a kernel running while that happens is slowed
        # down (measured: 8.6 instead of 3.6 ms for the first 4K frame): use both streams once now.
        for s in (self.render_stream, self.exchange_stream):
            torch.cuda.Event().record(s)
        torch.cuda.synchronize(self.dev)  # also waits for the zero fills, which ran on the caller's stream
        ctx.set_stream(self.render_stream.cuda_stream)  # rendering stays there; the untile names its stream per call

    @staticmethod
    def _ptr(t):
        return t.data_ptr() if t is not None else None

    def submit(self, params, timing=None):
        """Queues one frame; returns (slot, done_event).  The frame is `frame8[slot]` (and `frame[slot]`)
        once the event has completed, and stays valid until the second submit after this one.
        timing: optional pair of events recorded around the render kernel on the render stream."""
        b = self.count & 1
        if self.count >= 2:
            self.done[b].synchronize()  # at most two frames in flight: the host does not run ahead of the GPU without bound
        self.count += 1
        ctx, R, X = self.ctx, self.render_stream, self.exchange_stream
        ptr = self._ptr
        R.wait_event(self.released[b])  # recorded two frames ago (no-op the first time)
        if timing is not None:
            timing[0].record(R)
        ctx.render_shard_device(params, ptr(self.shard[b]), ptr(self.shard8[b]))
        if timing is not None:
            timing[1].record(R)
        self.rendered[b].record(R)
        X.wait_event(self.rendered[b])
        with torch.cuda.stream(X):
            src, src8 = self.shard[b], self.shard8[b]
            if self.gather:
                import torch.distributed as dist
                if self.shard[b] is not None:
                    dist.all_gather_into_tensor(self.gathered, self.shard[b], group=self.group)
                dist.all_gather_into_tensor(self.gathered8, self.shard8[b], group=self.group)
                self.released[b].record(X)
                src, src8 = self.gathered, self.gathered8
            ctx.untile_device_on(X.cuda_stream, params, ptr(src), ptr(self.frame[b]), ptr(src8), ptr(self.frame8[b]))
            if not self.gather:
                self.released[b].record(X)  # the untile read the shard itself
            self.done[b].record(X)
        return b, self.done[b]

    def drain(self):
        self.render_stream.synchronize()
        self.exchange_stream.synchronize()

    def close(self):
        """Waits for the queued frames and puts the context back on its own stream."""
        self.drain()
        self.ctx.set_stream(None)
