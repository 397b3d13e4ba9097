"""Where the host time of FramePipeline.submit goes (cProfile, tiny frame so the GPU never limits)."""
import cProfile, os, pstats, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
import torch
import ray_tracer_amd as rt
from ray_tracer_amd.frames import FramePipeline

ctx = rt.Context(0)
ctx.upload(rt.HostScene("teapot"))
w, h = 64, 64
p = ctx.params(w, h, mode=rt.MODE_GRID, grid_m=1)
pipe = FramePipeline(ctx, w, h)
for _ in range(50):
    pipe.submit(p)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(1000):
    pipe.submit(p)
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(14)
# the same without the bound on frames in flight
t0 = time.perf_counter()
for _ in range(1000):
    pipe.count = 0
    pipe.submit(p)
t1 = time.perf_counter()
torch.cuda.synchronize()
print(f"without the in-flight bound: host {1e3 * (t1 - t0):.1f} us per submit")
pipe.close()
ctx.close()
