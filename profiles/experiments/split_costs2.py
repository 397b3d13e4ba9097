"""Frame time of the teapot room with the mesh out of every ray's way (same kernel variant), per light count."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import ray_tracer_amd as rt
from split_costs import timeit

def room(pos, lights=2, planes=5):
    s = rt.HostScene()
    red = rt.Material(ambient=(0.01, 0.02, 0.02), diffuse=(1.0, 0.2, 0.2), specular=0.5, shininess=10.0)
    plain = rt.Material()
    s.add_mesh_obj(os.path.join(rt.ASSETS, "teapot.obj"), rt.translate(pos), red, boxed=True, flip_normals=True)
    walls = [((0, -3, 0), (0, 1, 0)), ((0, 1, 30), (0, 0, -1)), ((-15, 1, 0), (1, 0, 0)), ((15, 1, 0), (-1, 0, 0)), ((0, 27, 0), (0, -1, 0))]
    for p, n in walls[:planes]:
        s.add_plane(p, n, plain)
    for l in [((10, 16, -7), .5), ((2, 7, 1), .2)][:lights]:
        s.add_light(*l)
    return s

for lights in (2, 1, 0):
    timeit(f"teapot in view, {lights} lights", room((0, 1, 10), lights))
    timeit(f"teapot behind camera, {lights} lights", room((0, 1, -1000), lights))
timeit("teapot behind camera, 1 plane, 0 lights", room((0, 1, -1000), 0, 1))
