"""What the parts of a wall sample cost, by leaving them out at run time (no special build): the teapot preset's room
(five axis-aligned walls) with a one-triangle mesh far outside the view, so that the mesh kernel k_render<F_MESH> runs
and every ray goes primary query -> resolve -> one shading pass per light, none with a traced shadow ray.
Variants: 2 lights, 1 light, no light; then the teapot preset itself, with and without its lights.  One 4K / 16-tap frame each after a warm-up frame;
run under `rocprofv3 --pmc ...` the dispatches appear in this order (2 per variant), see profiles/run_phase_costs.sh."""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
import numpy as np
import ray_tracer_amd as rt

M = rt.Material
green = M((0.03, 0.1, 0.03), (0.3, 1.0, 0.3)); red = M((0.09, 0.06, 0.06), (0.9, 0.6, 0.6))
blue = M((0.06, 0.06, 0.09), (0.6, 0.6, 0.9)); plain = M()


def room(n_lights):
    sc = rt.HostScene()
    sc.add_plane((0, -3, 0), (0, 1, 0), plain); sc.add_plane((0, 1, 30), (0, 0, -1), green)
    sc.add_plane((-15, 1, 0), (1, 0, 0), red); sc.add_plane((15, 1, 0), (-1, 0, 0), blue); sc.add_plane((0, 27, 0), (0, -1, 0), plain)
    tri = np.array([[0, 100, -50, 1, 100, -50, 0, 101, -50]], np.float32)  # above the ceiling, behind the camera
    sc.add_mesh_tris(tri, np.tile(np.array([0, 0, 1], np.float32), (1, 3)), rt.mat_identity(), plain, boxed=False)
    for pos, c in [((10, 16, -7), .5), ((2, 7, 1), .2)][:n_lights]:
        sc.add_light(pos, c)
    return sc


class teapot_without_lights:
    """The teapot preset's scene description with the light count set to zero (the host scene stays alive with it)."""
    def __init__(self):
        self.scene = rt.HostScene("teapot")
        self.desc = self.scene.flatten()
        self.desc.contents.n_lights = 0


for name, sc in [("room_2_lights", room(2)), ("room_1_light", room(1)), ("room_0_lights", room(0)), ("teapot", rt.HostScene("teapot")),
                 ("teapot_0_lights", teapot_without_lights())]:
    ctx = rt.Context(0)
    ctx.upload(sc.desc if isinstance(sc, teapot_without_lights) else sc)
    ts = []
    for _ in range(2):
        ctx.render(3840, 2160, want_rgb=False, want_rgb8=True, mode=rt.MODE_GRID, grid_m=4, kernel=rt.KERNEL_PIXEL)
        ts.append(ctx.counters().last_trace_ms)
    c = ctx.counters()
    print(f"{name}: kernel ms {ts[-1]:.3f}, {rt.KERNEL_NAMES[c.render_variant]}, rays/frame {c.rays // 2}", flush=True)
    ctx.close()
