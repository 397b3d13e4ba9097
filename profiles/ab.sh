#!/bin/bash
# A/B of librt_amd.so builds on one GPU box, builds interleaved, REPS times over (paths relative to the repo root):
#   [WORKLOADS="teapot_4k_16spp shells_4k_64spp"] [STEPS=40] [REPS=2] bash profiles/ab.sh libA.so libB.so ...
# Defaults: the headline workload, 8 timed frames, two rounds.
cd "$GRAFT_REPO_ROOT"
for rep in $(seq 1 ${REPS:-2}); do
for w in ${WORKLOADS:-teapot_4k_16spp}; do
for lib in "$@"; do
    RT_AMD_LIB=$PWD/$lib timeout -k 10 200 python3 bench.py --steps ${STEPS:-8} --warmup 2 --no-cpu-baseline --workload $w 2>/dev/null | python3 -c "
import sys, json
j = json.loads(sys.stdin.readline())
print('$w', '$lib', 'ms/frame %.3f' % j['ms_per_step'], 'kernel %.3f' % j['roofline']['kernel_ms'], 'Mrays/s %.0f' % j['value'])" || exit 1
done
done
done
