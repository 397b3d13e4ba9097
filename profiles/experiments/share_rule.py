"""Slowest rank's kernel time of the 4K teapot frame at 2, 4, 8 ranks (all on this GPU) for the per-tile kernel and the
persistent queue kernel: which one the rule for short launches should pick."""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
import torch
import ray_tracer_amd as rt

w, h = 3840, 2160
buf = torch.zeros(w * h * 3 + 65536, dtype=torch.uint8, device="cuda")
ctx = rt.Context(0)
ctx.upload(rt.HostScene(sys.argv[1] if len(sys.argv) > 1 else "teapot"))
s = torch.cuda.Stream()
ctx.set_stream(s.cuda_stream)
for name, kernel in (("per-tile", rt.KERNEL_PER_TILE), ("queue", rt.KERNEL_QUEUE), ("rule", rt.KERNEL_AUTO)):
    for world in (1, 2, 4, 8):
        ts = []
        for rank in range(world):
            p = ctx.params(w, h, mode=rt.MODE_GRID, grid_m=4, rank=rank, world_size=world, kernel=kernel)
            for _ in range(3):
                ctx.render_shard_device(p, None, buf.data_ptr()); torch.cuda.synchronize()
            evs = []
            for _ in range(5):
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record(s); ctx.render_shard_device(p, None, buf.data_ptr()); b.record(s); evs.append((a, b))
            torch.cuda.synchronize()
            ts.append(min(a.elapsed_time(b) for a, b in evs))
        print(f"{name:9s} world {world}: slowest {max(ts):.3f} fastest {min(ts):.3f} ms ({rt.KERNEL_NAMES[ctx.counters().render_variant]})", flush=True)
ctx.set_stream(None); ctx.close()
