"""Render-kernel time of each rank's share of the 4K teapot frame for world sizes 1, 2, 4, 8 (all on this GPU)."""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
import torch
import ray_tracer_amd as rt

ctx = rt.Context(0)
ctx.upload(rt.HostScene(sys.argv[1] if len(sys.argv) > 1 else "teapot"))
w, h = 3840, 2160
s = torch.cuda.Stream()
ctx.set_stream(s.cuda_stream)
buf = torch.zeros(w * h * 3 + 65536, dtype=torch.uint8, device="cuda")
torch.cuda.synchronize()
for world in (1, 2, 4, 8):
    times, variants = [], []
    for rank in range(world):
        p = ctx.params(w, h, mode=rt.MODE_GRID, grid_m=4, rank=rank, world_size=world)
        for _ in range(3):  # warm-up; the per-tile kernel also needs one frame of tile times
            ctx.render_shard_device(p, None, buf.data_ptr())
            torch.cuda.synchronize()
        evs = []
        for _ in range(5):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(s); ctx.render_shard_device(p, None, buf.data_ptr()); b.record(s)
            evs.append((a, b))
        torch.cuda.synchronize()
        times.append(min(a.elapsed_time(b) for a, b in evs))
        variants.append(ctx.counters().render_variant)
    print(f"world {world}: " + " ".join("%.3f" % t for t in times) + f"   max {max(times):.3f} ms, kernel variants {variants}", flush=True)
ctx.set_stream(None)
ctx.close()
