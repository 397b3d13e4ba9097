#!/bin/bash
# A/B of librt_amd.so builds on one GPU box: bash profiles/ab.sh libA.so libB.so ...  (paths relative to the repo root)
cd "$GRAFT_REPO_ROOT"
for round in 1 2; do
for lib in "$@"; do
    RT_AMD_LIB=$PWD/$lib timeout -k 10 120 python3 bench.py --steps 8 --warmup 2 --no-cpu-baseline 2>/dev/null | python3 -c "
import sys, json
j = json.loads(sys.stdin.readline())
print('$lib', 'ms/frame %.3f' % j['ms_per_step'], 'kernel_ms %.3f' % j['roofline']['kernel_ms'], 'Mrays/s %.0f' % j['value'])" || exit 1
done
done
