"""Duration of the first render launches after an upload, by where the output buffers come from."""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
import torch
import ray_tracer_amd as rt

ctx = rt.Context(0)
ctx.upload(rt.HostScene("teapot"))
w, h = 3840, 2160
p = ctx.params(w, h, mode=rt.MODE_GRID, grid_m=4)
per, total = ctx.shard_tiles(p)
n = per * 256 * 3
s = torch.cuda.Stream()
ctx.set_stream(s.cuda_stream)

def run(tag, buf):
    evs = []
    for k in range(4):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(s); ctx.render_shard_device(p, None, buf.data_ptr()); b.record(s)
        evs.append((a, b))
    torch.cuda.synchronize()
    print(tag, " ".join("%.2f" % a.elapsed_time(b) for a, b in evs), flush=True)

b0 = torch.zeros(n, dtype=torch.uint8, device="cuda"); torch.cuda.synchronize()
run("zeros buffer, first use ", b0)
run("same buffer again       ", b0)
b1 = torch.empty(n, dtype=torch.uint8, device="cuda"); torch.cuda.synchronize()
run("empty buffer, first use ", b1)
b2 = torch.zeros(n, dtype=torch.uint8, device="cuda"); torch.cuda.synchronize()
run("second zeros buffer     ", b2)
big = [torch.zeros(n, dtype=torch.uint8, device="cuda") for _ in range(6)]; torch.cuda.synchronize()
run("after 6 more allocations", big[5])
ctx.set_stream(None)
ctx.close()
