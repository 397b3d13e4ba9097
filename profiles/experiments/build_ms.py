import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import ray_tracer_amd as rt
for preset in ("teapot", "shells"):
    best = 1e9
    for _ in range(4):
        ctx = rt.Context(0); ctx.upload(rt.HostScene(preset)); best = min(best, ctx.counters().last_build_ms); ctx.close()
    print(preset, "build ms %.3f" % best)
