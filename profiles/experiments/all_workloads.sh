#!/bin/bash
# bash profiles/experiments/all_workloads.sh [steps]: one bench line per workload (counts, executed rays, lanes)
cd "$GRAFT_REPO_ROOT"
steps=${1:-3}
for wl in analytic_4k_aa9 shells_4k_64spp textured_8k_256spp; do
    timeout -k 10 280 python3 bench.py --steps $steps --warmup 1 --no-cpu-baseline --workload $wl 2>/dev/null > gpurun_out/wl_$wl.json || exit 1
    python3 -c "
import json
j = json.loads(open('gpurun_out/wl_$wl.json').readline())
r = j['roofline']
print('$wl', r['kernel'], 'packet fill', r.get('packet_fill'), 'node steps (waves)', r['counts_per_launch'].get('node_steps_waves'), 'ms/frame %.3f' % j['ms_per_step'], 'kernel_ms %.3f' % r['kernel_ms'], 'Mrays/s %.0f' % j['value'], 'executed %.0f' % j['rays']['executed_Mrays_per_s'], 'frac %.4f' % r['frac'])"
done
