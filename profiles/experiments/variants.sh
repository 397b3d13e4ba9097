#!/bin/bash
# One bench workload under each render kernel: bash profiles/experiments/variants.sh <workload>
cd "$GRAFT_REPO_ROOT"
for v in 0 1 2 3; do
    RT_RENDER_VARIANT=$v timeout -k 10 200 python3 bench.py --steps 4 --warmup 3 --no-cpu-baseline --workload $1 2>/dev/null | python3 -c "
import sys, json
j = json.loads(sys.stdin.readline())
print('$1 variant $v', j['config']['render_kernel'], 'ms/frame %.3f' % j['ms_per_step'], 'kernel %.3f' % j['roofline']['kernel_ms'])" || exit 1
done
