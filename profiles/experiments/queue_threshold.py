"""One rank's share of the 4K teapot frame by the queue kernel, by world size and forced heavy-tile threshold (10 ns ticks;
0: the kernel's own rule):  python3 profiles/experiments/queue_threshold.py"""
import os, subprocess, sys
if len(sys.argv) < 2:
    for th in (0, 1000, 2000, 4000, 8000, 16000, 4000000000):
        env = dict(os.environ)
        if th: env["RT_HYBRID_THRESHOLD"] = str(th)
        subprocess.run([sys.executable, __file__, str(th)], env=env, check=True)
    sys.exit(0)
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
import torch
import ray_tracer_amd as rt
w, h = 3840, 2160
buf = torch.zeros(w * h * 3 + 65536, dtype=torch.uint8, device="cuda")
ctx = rt.Context(0)
ctx.upload(rt.HostScene("teapot"))
s = torch.cuda.Stream()
ctx.set_stream(s.cuda_stream)
out = []
for world in (1, 2, 4, 8):
    p = ctx.params(w, h, mode=rt.MODE_GRID, grid_m=4, rank=0, world_size=world, kernel=rt.KERNEL_QUEUE)
    for _ in range(3):
        ctx.render_shard_device(p, None, buf.data_ptr()); torch.cuda.synchronize()
    evs = []
    for _ in range(6):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(s); ctx.render_shard_device(p, None, buf.data_ptr()); b.record(s); evs.append((a, b))
    torch.cuda.synchronize()
    out.append("world %d: %.3f" % (world, min(a.elapsed_time(b) for a, b in evs)))
print("threshold", sys.argv[1], " ".join(out), flush=True)
ctx.set_stream(None); ctx.close()
