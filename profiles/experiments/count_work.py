import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import ray_tracer_amd as rt
ctx = rt.Context(0); ctx.upload(rt.HostScene("teapot"))
p = dict(mode=rt.MODE_GRID, grid_m=4)
ctx.render(3840, 2160, want_rgb=False, want_rgb8=True, **p)
ctx.reset_counters()
ctx.render(3840, 2160, want_rgb=False, want_rgb8=True, count_work=True, **p)
c = ctx.counters()
print({k: getattr(c, k) for k, _ in c._fields_})
