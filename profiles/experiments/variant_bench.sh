#!/bin/bash
# bash profiles/experiments/variant_bench.sh lib.so "0 4" [workload]: bench of one library build with the render kernel forced to each listed variant
cd "$GRAFT_REPO_ROOT"
lib=$1; variants=$2; wl=${3:-teapot_4k_16spp}
for v in $variants; do
    RT_RENDER_VARIANT=$v RT_AMD_LIB=$PWD/$lib timeout -k 10 150 python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --workload $wl 2>/dev/null | python3 -c "
import sys, json
j = json.loads(sys.stdin.readline())
print('$lib variant $v', j['config']['render_kernel'], 'ms/frame %.3f' % j['ms_per_step'], 'kernel_ms %.3f' % j['roofline']['kernel_ms'], 'Mrays/s %.0f' % j['value'])" || exit 1
done
