import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import ray_tracer_amd as rt
for _ in range(2):
    ctx = rt.Context(0); ctx.upload(rt.HostScene("shells")); print("build ms", ctx.counters().last_build_ms); ctx.close()
