#!/bin/bash
# The two instruction-count passes of run_pmc.sh alone (a quick look at what a change did): bash profiles/run_pmc_quick.sh <tag>
tag=${1:-quick}
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/pmc_$tag
run() {
    name=$1; shift
    timeout -s KILL 120 rocprofv3 --pmc "$@" --output-format csv -d gpurun_out/pmc_$tag/$name -o pmc -- \
        python3 bench.py --steps 2 --warmup 3 --no-cpu-baseline > gpurun_out/pmc_$tag/$name.log 2>&1 || { echo "pass $name failed"; tail -5 gpurun_out/pmc_$tag/$name.log; return 1; }
    echo "pass $name ok"
}
run sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY || exit 1
run sq2 SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS || exit 1
python3 profiles/summarise_pmc.py gpurun_out/pmc_$tag "k_render<1, false>" > gpurun_out/pmc_$tag/summary.json
