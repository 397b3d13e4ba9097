#!/bin/bash
# Rank-share kernel times under the timing-based choice and under each forced kernel.
cd "$GRAFT_REPO_ROOT"
echo "--- chosen by timing"; timeout -k 10 120 python3 profiles/experiments/rank_share.py 2>/dev/null || exit 1
for v in 1 2 3; do
echo "--- forced variant $v"; RT_RENDER_VARIANT=$v timeout -k 10 120 python3 profiles/experiments/rank_share.py 2>/dev/null || exit 1
done
