"""Pure traversal throughput: a close-up of the teapot with no walls and no lights (run on a GPU box)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import ray_tracer_amd as rt

def scene(lights):
    s = rt.HostScene()
    red = rt.Material(ambient=(0.01, 0.02, 0.02), diffuse=(1.0, 0.2, 0.2), specular=0.5, shininess=10.0)
    s.add_mesh_obj(os.path.join(rt.ASSETS, "teapot.obj"), rt.translate((0, -1.6, 4.2)), red, boxed=True, flip_normals=True)
    for l in [((10, 16, -7), .5), ((2, 7, 1), .2)][:lights]:
        s.add_light(*l)
    return s

for lights in (0, 1):
    ctx = rt.Context(0)
    ctx.upload(scene(lights))
    w, h = 3840, 2160
    for count in (False, True):
        for _ in range(2):
            ctx.render(w, h, want_rgb=False, mode=rt.MODE_GRID, grid_m=2, count_work=count)
        ctx.reset_counters()
        ctx.render(w, h, want_rgb=False, mode=rt.MODE_GRID, grid_m=2, count_work=count)
        c = ctx.counters()
        print(f"lights={lights} count={count}: kernel {c.last_trace_ms:.3f} ms rays {c.rays/1e6:.1f}M nodes {c.box_tests/2e6:.1f}M tris {c.tri_tests/1e6:.1f}M gates {c.gate_tests/1e6:.1f}M"
              + (f"  -> {c.last_trace_ms*1e6/(c.box_tests/2+1):.3f} ns/node" if count else ""))
    ctx.close()
