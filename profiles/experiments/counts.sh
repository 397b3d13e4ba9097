#!/bin/bash
# bash profiles/experiments/counts.sh lib.so ... : work counters of the headline frame for each library build
cd "$GRAFT_REPO_ROOT"
for lib in "$@"; do
    RT_AMD_LIB=$PWD/$lib timeout -k 10 150 python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python3 -c "
import sys, json
j = json.loads(sys.stdin.readline())
c = j['roofline']['counts_per_launch']
print('$lib', 'kernel_ms %.3f' % j['roofline']['kernel_ms'], {k: round(v / 1e6, 1) for k, v in c.items()}, 'shadow queries M', j['rays']['shadow_queries_executed_per_frame'] / 1e6)" || exit 1
done
