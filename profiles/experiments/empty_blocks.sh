# What the blocks of the per-tile kernel that leave at once cost: the headline frame with every tile light (three of a
# tile's four blocks leave) beside the one-block-per-tile kernel, and with every tile heavy.
#   bash profiles/experiments/empty_blocks.sh
cd $GRAFT_REPO_ROOT
run() {
  timeout -k 10 100 python3 bench.py --steps 30 --warmup 6 --no-cpu-baseline 2>/dev/null | python3 -c "
import sys,json
j=json.loads(sys.stdin.readline()); print('$1', 'ms %.4f' % j['ms_per_step'], 'kernel %.4f' % j['roofline']['kernel_ms'], j['roofline']['kernel'])" || exit 1
}
RT_RENDER_VARIANT=0 run "one block per tile"
RT_RENDER_VARIANT=3 RT_HYBRID_THRESHOLD=4000000000 run "per-tile kernel, all tiles light"
RT_RENDER_VARIANT=3 run "per-tile kernel, threshold 1.5 x mean"
RT_RENDER_VARIANT=3 RT_HYBRID_THRESHOLD=0 run "per-tile kernel, all tiles heavy"
RT_RENDER_VARIANT=1 run "every tile by four blocks"
