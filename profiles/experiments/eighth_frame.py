"""One rank's share of the 4K frame at 8 GPUs, emulated on one GPU by a frame of 1/8 the tiles
(1360x768, same scene and taps): render-kernel time and end-to-end time per pipelined frame."""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
import torch
import torch.distributed as dist
import ray_tracer_amd as rt
from ray_tracer_amd.frames import FramePipeline

ctx = rt.Context(0)
ctx.upload(rt.HostScene("teapot"))
w, h = 1360, 768
p = ctx.params(w, h, mode=rt.MODE_GRID, grid_m=4)
dist.init_process_group("nccl", init_method="tcp://127.0.0.1:29556", rank=0, world_size=1, device_id=torch.device("cuda", 0))
for coll in (False, True):
    pipe = FramePipeline(ctx, w, h, force_collective=coll)
    for _ in range(20):
        pipe.submit(p)
    torch.cuda.synchronize()
    n = 500
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    t0 = time.perf_counter()
    for k in range(n):
        pipe.submit(p, evs[k])
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    km = sum(a.elapsed_time(b) for a, b in evs) / n
    print(f"collective={coll}: kernel {km * 1e3:.0f} us, host {1e6 * (t1 - t0) / n:.0f} us per submit, {1e6 * (t2 - t0) / n:.0f} us per frame end to end")
    pipe.close()
dist.destroy_process_group()
ctx.close()
