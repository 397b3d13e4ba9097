"""Renders a preset a few times (for profiling): python render_once.py preset width height grid_m [frames]"""
import os
import sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
import ray_tracer_amd as rt

preset, w, h, m = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
frames = int(sys.argv[5]) if len(sys.argv) > 5 else 2
ctx = rt.Context(0)
ctx.upload(rt.HostScene(preset))
for _ in range(frames):
    ctx.render(w, h, want_rgb=False, want_rgb8=True, mode=rt.MODE_GRID, grid_m=m)
c = ctx.counters()
print(preset, "kernel ms", c.last_trace_ms, "rays/frame", c.rays / frames)
