"""Renders a bench workload a few times at a reduced frame size (for counter passes that would take
too long at full size): python profiles/experiments/render_workload.py <workload> <divisor> [frames]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
import ray_tracer_amd as rt  # noqa: E402

name, div = sys.argv[1], int(sys.argv[2])
frames = int(sys.argv[3]) if len(sys.argv) > 3 else 3
preset, w, h, mode, grid_m, depth = bench.WORKLOADS[name]
w, h = w // div, h // div
ctx = rt.Context(0)
ctx.upload(rt.HostScene(preset))
p = ctx.params(w, h, mode=mode, grid_m=grid_m, max_depth=depth)
per_rank, _ = ctx.shard_tiles(p)
shard8 = torch.zeros(per_rank * rt.TILE * rt.TILE * 3, dtype=torch.uint8, device="cuda")
for k in range(frames):
    ctx.render_shard_device(p, None, shard8.data_ptr())
    torch.cuda.synchronize()
    c = ctx.counters()
    print(f"{name} {w}x{h} frame {k}: {c.last_trace_ms:.2f} ms, kernel variant {c.render_variant}", flush=True)
ctx.close()
