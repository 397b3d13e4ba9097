"""Host time per FramePipeline.submit with a frame so small that the GPU never limits: with and without the all-gather."""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
import torch
import torch.distributed as dist
import ray_tracer_amd as rt
from ray_tracer_amd.frames import FramePipeline

ctx = rt.Context(0)
ctx.upload(rt.HostScene("teapot"))
w, h = 64, 64
p = ctx.params(w, h, mode=rt.MODE_GRID, grid_m=1)
dist.init_process_group("nccl", init_method="tcp://127.0.0.1:29555", rank=0, world_size=1, device_id=torch.device("cuda", 0))
for coll in (False, True):
    pipe = FramePipeline(ctx, w, h, force_collective=coll)
    for _ in range(50):
        pipe.submit(p)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 2000
    for _ in range(n):
        pipe.submit(p)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"collective={coll}: host {1e6 * (t1 - t0) / n:.1f} us per submit, {1e6 * (t2 - t0) / n:.1f} us per frame end to end")
    pipe.close()
dist.destroy_process_group()
ctx.close()
