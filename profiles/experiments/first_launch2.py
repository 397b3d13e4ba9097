"""Render-kernel duration of the first frames through FramePipeline."""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
import torch
import ray_tracer_amd as rt
from ray_tracer_amd.frames import FramePipeline

ctx = rt.Context(0)
ctx.upload(rt.HostScene("teapot"))
w, h = 3840, 2160
p = ctx.params(w, h, mode=rt.MODE_GRID, grid_m=4)
pipe = FramePipeline(ctx, w, h)
evs = []
for k in range(6):
    t = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
    pipe.submit(p, t)
    evs.append(t)
torch.cuda.synchronize()
print(" ".join("%.2f" % a.elapsed_time(b) for a, b in evs))
pipe.close()
ctx.close()
