"""Where the frame time goes: the teapot room without the mesh, with fewer lights, etc. (run on a GPU box)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import ray_tracer_amd as rt

def room(mesh=True, lights=2, planes=5):
    s = rt.HostScene()
    red = rt.Material(ambient=(0.01, 0.02, 0.02), diffuse=(1.0, 0.2, 0.2), specular=0.5, shininess=10.0)
    plain = rt.Material()
    if mesh:
        s.add_mesh_obj(os.path.join(rt.ASSETS, "teapot.obj"), rt.translate((0, 1, 10)), red, boxed=True, flip_normals=True)
    walls = [((0, -3, 0), (0, 1, 0)), ((0, 1, 30), (0, 0, -1)), ((-15, 1, 0), (1, 0, 0)), ((15, 1, 0), (-1, 0, 0)), ((0, 27, 0), (0, -1, 0))]
    for p, n in walls[:planes]:
        s.add_plane(p, n, plain)
    for l in [((10, 16, -7), .5), ((2, 7, 1), .2)][:lights]:
        s.add_light(*l)
    return s

def timeit(name, scene, w=3840, h=2160, grid=4):
    ctx = rt.Context(0)
    ctx.upload(scene)
    for _ in range(2):
        ctx.render(w, h, want_rgb=False, want_rgb8=True, mode=rt.MODE_GRID, grid_m=grid)
    ctx.reset_counters()
    ctx.render(w, h, want_rgb=False, want_rgb8=True, mode=rt.MODE_GRID, grid_m=grid)
    c = ctx.counters()
    print(f"{name:32s} kernel {c.last_trace_ms:7.3f} ms  rays {c.rays/1e6:7.1f} M  {c.rays/c.last_trace_ms/1e3:8.0f} Mrays/s")
    ctx.close()

if __name__ == "__main__":
    timeit("room+teapot, 2 lights", room())
    timeit("room only, 2 lights", room(mesh=False))
    timeit("room+teapot, 1 light", room(lights=1))
    timeit("room only, 1 light", room(mesh=False, lights=1))
    timeit("room+teapot, 0 lights", room(lights=0))
    timeit("room only, 0 lights", room(mesh=False, lights=0))
    timeit("floor+teapot, 2 lights", room(planes=1))
