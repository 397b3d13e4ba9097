"""End-to-end time per pipelined frame when this GPU renders only rank 0's share of the 4K teapot frame
for world sizes 1..8 (the exchange is a single-rank all-gather of the whole frame buffer, i.e. a local copy
of the bytes a real all-gather would deliver; the untiled picture is meaningless, the timing is not)."""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
import torch
import torch.distributed as dist
import ray_tracer_amd as rt
from ray_tracer_amd.frames import FramePipeline

ctx = rt.Context(0)
ctx.upload(rt.HostScene("teapot"))
w, h = 3840, 2160
dist.init_process_group("nccl", init_method="tcp://127.0.0.1:29557", rank=0, world_size=1, device_id=torch.device("cuda", 0))
for world in (1, 2, 4, 8):
    p = ctx.params(w, h, mode=rt.MODE_GRID, grid_m=4, rank=0, world_size=world)
    pipe = FramePipeline(ctx, w, h, force_collective=True)
    for _ in range(4):
        pipe.submit(p)
    torch.cuda.synchronize()
    n = 40
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    t0 = time.perf_counter()
    for k in range(n):
        pipe.submit(p, evs[k])
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    km = sum(a.elapsed_time(b) for a, b in evs) / n
    print(f"world {world}: kernel {km:.3f} ms, end to end {1e3 * (t2 - t0) / n:.3f} ms per frame, variant {ctx.counters().render_variant}", flush=True)
    pipe.close()
dist.destroy_process_group()
ctx.close()
