#!/bin/bash
# Per-kernel timing of the headline bench with rocprofv3 (run on the GPU box from the repo root):
#   bash profiles/run_kernel_trace.sh <tag>
# Writes gpurun_out/prof_<tag>/ ; copy the *_kernel_stats.csv into profiles/ to keep it.
set -e
tag=${1:-r01}
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$tag -o bench -- \
    python3 bench.py --steps 5 --warmup 3 --no-cpu-baseline > gpurun_out/prof_${tag}_bench.log 2>&1
find gpurun_out/prof_$tag -name '*stats*' | head
