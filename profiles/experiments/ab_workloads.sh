#!/bin/bash
# bash profiles/experiments/ab_workloads.sh "libA.so libB.so" "workload ..." [variant]: frame time of each build on each workload
cd "$GRAFT_REPO_ROOT"
for wl in $2; do for lib in $1; do
    ${3:+env RT_RENDER_VARIANT=$3} env RT_AMD_LIB=$PWD/$lib timeout -k 10 200 python3 bench.py --steps ${STEPS:-5} --warmup ${WARMUP:-2} --no-cpu-baseline --workload $wl 2>/dev/null | python3 -c "
import sys, json
j = json.loads(sys.stdin.readline())
print('$wl $lib', j['config']['render_kernel'], 'ms/frame %.3f' % j['ms_per_step'], 'gate_tri %.1fM' % (j['roofline']['counts_per_launch']['gate_tri_tests'] / 1e6))" || exit 1
done; done
