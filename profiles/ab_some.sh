#!/bin/bash
# Chosen bench workloads for one or more builds of librt_amd.so:
#   WORKLOADS="analytic_4k_aa9 textured_8k_256spp" bash profiles/ab_some.sh libA.so [libB.so ...]
cd "$GRAFT_REPO_ROOT"
for w in $WORKLOADS; do
for lib in "$@"; do
    RT_AMD_LIB=$PWD/$lib timeout -k 10 200 python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --workload $w 2>/dev/null | python3 -c "
import sys, json
j = json.loads(sys.stdin.readline())
print('$w', '$lib', 'ms/frame %.3f' % j['ms_per_step'], 'Mrays/s %.0f' % j['value'])" || exit 1
done
done
