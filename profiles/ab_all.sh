#!/bin/bash
# All bench workloads for one or more builds of librt_amd.so: bash profiles/ab_all.sh libA.so [libB.so ...]
cd "$GRAFT_REPO_ROOT"
for w in teapot_4k_16spp analytic_4k_aa9 shells_4k_64spp textured_8k_256spp; do
for lib in "$@"; do
    RT_AMD_LIB=$PWD/$lib timeout -k 10 200 python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --workload $w 2>/dev/null | python3 -c "
import sys, json
j = json.loads(sys.stdin.readline())
print('$w', '$lib', 'ms/frame %.3f' % j['ms_per_step'], 'Mrays/s %.0f' % j['value'])" || exit 1
done
done
