import os, sys
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import ray_tracer_amd as rt
ctx = rt.Context(0); ctx.upload(rt.HostScene("teapot"))
ctx.reset_counters()
ctx.render(3840, 2160, want_rgb=False, mode=rt.MODE_GRID, grid_m=4, count_work=True)
c = ctx.counters()
print(os.environ.get("RT_AMD_LIB","").split("/")[-1], "node steps (waves) %.2fM leaf steps (waves) %.2fM  lane box tests %.0fM lane tri tests %.1fM" % (c.tex_fetches/1e6, c.quadric_tests/1e6, c.box_tests/1e6, c.tri_tests/1e6))
