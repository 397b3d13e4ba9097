# Headline frame time of the per-tile kernel by the heavy-tile threshold (10 ns ticks): bash profiles/experiments/threshold_sweep.sh
cd $GRAFT_REPO_ROOT
for th in 4000 6000 8000 10000 12000 16000 24000 40000; do
  RT_HYBRID_THRESHOLD=$th timeout -k 10 100 python3 bench.py --steps 30 --warmup 6 --no-cpu-baseline 2>/dev/null | python3 -c "
import sys,json
j=json.loads(sys.stdin.readline()); print('threshold $th', 'ms %.4f' % j['ms_per_step'], j['roofline']['kernel'])" || exit 1
done
