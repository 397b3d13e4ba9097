# The scenes without meshes by k_render (the rule) and by the queue kernel: bash profiles/experiments/queue_no_mesh.sh
cd $GRAFT_REPO_ROOT
for spec in "analytic_4k_aa9 20" "textured_8k_256spp 6"; do
set -- $spec
for v in 0 4; do
  RT_RENDER_VARIANT=$v timeout -k 10 200 python3 bench.py --workload $1 --steps $2 --warmup 3 --no-cpu-baseline 2>/dev/null | python3 -c "
import sys,json
j=json.loads(sys.stdin.readline()); print('$1 variant $v', 'ms %.4f' % j['ms_per_step'], 'kernel %.4f' % j['roofline']['kernel_ms'], j['roofline']['kernel'])" || exit 1
done
done
