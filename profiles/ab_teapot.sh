#!/bin/bash
# Headline workload, 40 timed frames per build, builds interleaved twice: bash profiles/ab_teapot.sh libA.so libB.so
cd "$GRAFT_REPO_ROOT"
for rep in 1 2; do
for lib in "$@"; do
    RT_AMD_LIB=$PWD/$lib timeout -k 10 200 python3 bench.py --steps 40 --warmup 5 --no-cpu-baseline 2>/dev/null | python3 -c "
import sys, json
j = json.loads(sys.stdin.readline())
print('$lib', 'ms/frame %.3f' % j['ms_per_step'], 'kernel %.3f' % j['roofline']['kernel_ms'])" || exit 1
done
done
