#!/bin/bash
# Everything the per-workload roofline lines are made of, in one go on the GPU box (from the repo root):
#   bash profiles/run_r03_evidence.sh <tag>
# 1. rocprofv3 --kernel-trace --stats of the headline bench; 2. counter passes (one rocprofv3 --pmc run each) of the
# headline at full size and of the other workloads (shells at half, textured at a quarter of width and height, scaled
# back by make_traffic.py); 3. profiles/hbm_traffic.json rebuilt from them; 4. one bench line per workload.
# Results under gpurun_out/evidence_<tag>/ ; copy what is to be kept into profiles/.
tag=${1:-r03}
out=gpurun_out/evidence_$tag
mkdir -p $out
bash profiles/run_kernel_trace.sh $tag || exit 1
cp $(find gpurun_out/prof_$tag -name '*kernel_stats.csv' | head -1) $out/kernel_stats.csv || exit 1
bash profiles/run_pmc.sh $tag || exit 1
mkdir -p profiles/$tag
python3 profiles/summarise_pmc.py gpurun_out/pmc_$tag "k_render_queue<1>" > profiles/$tag/pmc_teapot.json || exit 1
python3 profiles/make_traffic.py profiles/$tag/pmc_teapot.json teapot_4k_16spp k_render_queue > /dev/null || exit 1
for spec in analytic_4k_aa9:1:analytic:k_render shells_4k_64spp:2:shells_half:k_render_queue textured_8k_256spp:4:textured_quarter:k_render; do
    IFS=: read wl div name kernel <<< "$spec"
    bash profiles/run_pmc_workload.sh ${tag}_$name $wl $div || exit 1
    python3 profiles/summarise_pmc.py gpurun_out/pmc_${tag}_$name "k_render" > profiles/$tag/pmc_$name.json || exit 1
    python3 profiles/make_traffic.py profiles/$tag/pmc_$name.json $wl $kernel $((div * div)) > /dev/null || exit 1
done
cp profiles/hbm_traffic.json profiles/$tag/pmc_*.json $out/
bash profiles/run_bench_lines.sh evidence_$tag || exit 1
echo evidence done
