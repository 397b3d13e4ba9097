"""The kernels of one rank's share of the headline frame, for `rocprofv3 --kernel-trace --stats`:
python profiles/experiments/share_kernels.py <world_size> [frames]"""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
import torch
import ray_tracer_amd as rt

world = int(sys.argv[1]); frames = int(sys.argv[2]) if len(sys.argv) > 2 else 50
ctx = rt.Context(0)
ctx.upload(rt.HostScene("teapot"))
p = ctx.params(3840, 2160, mode=rt.MODE_GRID, grid_m=4, rank=0, world_size=world)
per_rank, _ = ctx.shard_tiles(p)
buf = torch.zeros(per_rank * rt.TILE * rt.TILE * 3, dtype=torch.uint8, device="cuda")
for _ in range(frames):
    ctx.render_shard_device(p, None, buf.data_ptr())
ctx.sync()
print("variant", ctx.counters().render_variant, "last kernel ms", ctx.counters().last_trace_ms)
ctx.close()
