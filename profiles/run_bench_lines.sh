#!/bin/bash
# One bench line per workload (the headline with its CPU baseline): bash profiles/run_bench_lines.sh <out dir under gpurun_out>
out=gpurun_out/${1:-bench_lines}
mkdir -p $out
timeout -k 10 200 python3 bench.py > $out/bench_teapot_4k_16spp.json 2> $out/bench_teapot.err || exit 1
timeout -k 10 120 python3 bench.py --workload analytic_4k_aa9 --no-cpu-baseline > $out/bench_analytic_4k_aa9.json 2> $out/bench_analytic.err || exit 1
timeout -k 10 120 python3 bench.py --workload shells_4k_64spp --no-cpu-baseline --steps 20 > $out/bench_shells_4k_64spp.json 2> $out/bench_shells.err || exit 1
timeout -k 10 200 python3 bench.py --workload textured_8k_256spp --no-cpu-baseline --steps 8 --warmup 2 > $out/bench_textured_8k_256spp.json 2> $out/bench_textured.err || exit 1
echo bench lines done
