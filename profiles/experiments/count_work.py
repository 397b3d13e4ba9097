"""Work counters of one frame: python count_work.py [preset] [width] [height] [grid_m]"""
import os
import sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
import ray_tracer_amd as rt

preset = sys.argv[1] if len(sys.argv) > 1 else "teapot"
w, h, m = (int(sys.argv[i]) if len(sys.argv) > i else v for i, v in ((2, 3840), (3, 2160), (4, 4)))
ctx = rt.Context(0)
ctx.upload(rt.HostScene(preset))
ctx.render(w, h, want_rgb=False, want_rgb8=True, mode=rt.MODE_GRID, grid_m=m)
ctx.reset_counters()
ctx.render(w, h, want_rgb=False, want_rgb8=True, mode=rt.MODE_GRID, grid_m=m, count_work=True)
c = ctx.counters()
d = {k: getattr(c, k) for k, _ in c._fields_}
print(d)
rays = d["rays_primary"] + d["rays_shadow"] + d["rays_secondary"]
print(f"per ray: tri {d['tri_tests'] / rays:.2f} box {d['box_tests'] / rays:.2f} gate {d['gate_tests'] / rays:.3f} plane {d['plane_tests'] / rays:.2f}")
