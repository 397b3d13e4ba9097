# Whole frames on one GPU by the per-tile kernel and by the queue kernel, mesh workloads: bash profiles/experiments/queue_vs_per_tile.sh
cd $GRAFT_REPO_ROOT
for w in teapot_4k_16spp shells_4k_64spp; do
for v in 3 4; do
  RT_RENDER_VARIANT=$v timeout -k 10 150 python3 bench.py --workload $w --steps 20 --warmup 4 --no-cpu-baseline 2>/dev/null | python3 -c "
import sys,json
j=json.loads(sys.stdin.readline()); print('$w variant $v', 'ms %.4f' % j['ms_per_step'], 'kernel %.4f' % j['roofline']['kernel_ms'], j['roofline']['kernel'])" || exit 1
done
done
