"""Kernel time of the teapot preset's room without the teapot (five axis-aligned walls, two lights), 4K, 16 taps:
what the plane-only part of the headline frame costs.  RT_AMD_LIB selects the library build."""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
import ray_tracer_amd as rt

M = rt.Material
green = M((0.03, 0.1, 0.03), (0.3, 1.0, 0.3)); red = M((0.09, 0.06, 0.06), (0.9, 0.6, 0.6))
blue = M((0.06, 0.06, 0.09), (0.6, 0.6, 0.9)); plain = M()
sc = rt.HostScene()
sc.add_plane((0, -3, 0), (0, 1, 0), plain); sc.add_plane((0, 1, 30), (0, 0, -1), green)
sc.add_plane((-15, 1, 0), (1, 0, 0), red); sc.add_plane((15, 1, 0), (-1, 0, 0), blue); sc.add_plane((0, 27, 0), (0, -1, 0), plain)
sc.add_light((10, 16, -7), .5); sc.add_light((2, 7, 1), .2)
ctx = rt.Context(0)
ctx.upload(sc)
ts = []
for _ in range(8):
    ctx.render(3840, 2160, want_rgb=False, want_rgb8=True, mode=rt.MODE_GRID, grid_m=4)
    ts.append(ctx.counters().last_trace_ms)
c = ctx.counters()
print("walls only", os.environ.get("RT_AMD_LIB", "default").split("/")[-1], "kernel ms min %.3f" % min(ts[1:]), "variant", c.render_variant, "rays/frame", c.rays // 8)
