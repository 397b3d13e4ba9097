"""Kernel time of rank 0's share of the 4K teapot frame at 8 ranks for the render kernels and for the per-tile kernel
with every tile light / heavy: where a short launch loses time.  (RT_RENDER_VARIANT / RT_HYBRID_THRESHOLD are read when
a context is created.)"""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
import torch
import ray_tracer_amd as rt

w, h = 3840, 2160
buf = torch.zeros(w * h * 3 + 65536, dtype=torch.uint8, device="cuda")
for name, env in [("k_render", {"RT_RENDER_VARIANT": "0"}), ("split<1>", {"RT_RENDER_VARIANT": "1"}),
                  ("per-tile, all light", {"RT_RENDER_VARIANT": "3", "RT_HYBRID_THRESHOLD": "4000000000"}),
                  ("per-tile, all heavy", {"RT_RENDER_VARIANT": "3", "RT_HYBRID_THRESHOLD": "0"}),
                  ("per-tile", {"RT_RENDER_VARIANT": "3"}), ("queue", {"RT_RENDER_VARIANT": "4"})]:
    for k in ("RT_RENDER_VARIANT", "RT_HYBRID_THRESHOLD"):
        os.environ.pop(k, None)
    os.environ.update(env)
    ctx = rt.Context(0)
    ctx.upload(rt.HostScene("teapot"))
    s = torch.cuda.Stream()
    ctx.set_stream(s.cuda_stream)
    out = []
    for world in (8, 1):
        p = ctx.params(w, h, mode=rt.MODE_GRID, grid_m=4, rank=0, world_size=world)
        for _ in range(3):
            ctx.render_shard_device(p, None, buf.data_ptr()); torch.cuda.synchronize()
        evs = []
        for _ in range(6):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(s); ctx.render_shard_device(p, None, buf.data_ptr()); b.record(s); evs.append((a, b))
        torch.cuda.synchronize()
        out.append(min(a.elapsed_time(b) for a, b in evs))
    print(f"{name:22s} rank 0 of 8: {out[0]:.3f} ms   whole frame: {out[1]:.3f} ms", flush=True)
    ctx.set_stream(None); ctx.close()
