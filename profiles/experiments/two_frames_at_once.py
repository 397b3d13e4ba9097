"""Is a frame's time its total work, or does part of the GPU idle while the last heavy blocks finish?  Two contexts render
the headline frame on streams of their own, first one after the other, then side by side: if the pair side by side takes
clearly less than twice one frame, a single launch leaves issue slots unused (tail, uneven blocks)."""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
import torch
import ray_tracer_amd as rt

W, H, M = 3840, 2160, 4
ctxs, bufs, ps = [], [], []
for k in range(2):
    c = rt.Context(0)
    c.upload(rt.HostScene("teapot"))
    p = c.params(W, H, mode=rt.MODE_GRID, grid_m=M)
    per_rank, _ = c.shard_tiles(p)
    ctxs.append(c); ps.append(p)
    bufs.append(torch.zeros(per_rank * rt.TILE * rt.TILE * 3, dtype=torch.uint8, device="cuda"))


def run(which, frames):
    for c in ctxs:
        c.sync()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(frames):
        for k in which:
            ctxs[k].render_shard_device(ps[k], None, bufs[k].data_ptr())
    for k in which:
        ctxs[k].sync()
    return (time.perf_counter() - t0) / frames * 1e3


for rep in range(2):
    run([0, 1], 5)
    one = run([0], 40)
    other = run([1], 40)
    both = run([0, 1], 40)
    print(f"one context {one:.3f} ms per frame, the other {other:.3f}; both side by side {both:.3f} ms per pair ({both / (one + other):.2f} of the sum)", flush=True)
