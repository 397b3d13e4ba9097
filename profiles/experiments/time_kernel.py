"""Kernel time of a preset from the context's own events: python time_kernel.py preset width height grid_m"""
import os
import sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
import ray_tracer_amd as rt

preset, w, h, m = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
ctx = rt.Context(0)
ctx.upload(rt.HostScene(preset))
ts = []
for _ in range(6):
    ctx.render(w, h, want_rgb=False, want_rgb8=True, mode=rt.MODE_GRID, grid_m=m)
    ts.append(ctx.counters().last_trace_ms)
print(preset, os.environ.get("RT_AMD_LIB", "default").split("/")[-1], "split>=", os.environ.get("RT_SPLIT_MIN_TAPS", "32"), "kernel ms min %.3f" % min(ts[1:]))
