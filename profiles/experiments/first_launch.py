"""First-frame render time through FramePipeline under variations: python first_launch.py <variant>"""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
import torch
import ray_tracer_amd as rt
from ray_tracer_amd import frames
from ray_tracer_amd.frames import FramePipeline

variant = sys.argv[1]
if variant == "empty":
    torch_zeros = torch.zeros
    torch.zeros = lambda *a, **k: torch.empty(*a, **k)
ctx = rt.Context(0)
ctx.upload(rt.HostScene("teapot"))
w, h = 3840, 2160
p = ctx.params(w, h, mode=rt.MODE_GRID, grid_m=4)
pipe = FramePipeline(ctx, w, h)
if variant == "touch":
    for s in (pipe.render_stream, pipe.exchange_stream):
        e = torch.cuda.Event(); e.record(s)
    torch.cuda.synchronize()
if variant == "sleep":
    import time; time.sleep(0.5)
evs = []
for k in range(4):
    t = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
    pipe.submit(p, t)
    evs.append(t)
torch.cuda.synchronize()
print(variant, " ".join("%.2f" % a.elapsed_time(b) for a, b in evs))
pipe.close()
ctx.close()
