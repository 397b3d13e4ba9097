#!/bin/bash
# Counters for an arbitrary python script: bash profiles/run_pmc_cmd.sh <tag> <script.py> [args]
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/pmc_$tag
run() {
    name=$1; shift
    timeout -s KILL 120 rocprofv3 --pmc $PMC --output-format csv -d gpurun_out/pmc_$tag/$name -o pmc -- python3 "$@" > gpurun_out/pmc_$tag/$name.log 2>&1 || { echo "pass $name failed"; tail -5 gpurun_out/pmc_$tag/$name.log; return 1; }
    echo "pass $name ok"
}
PMC="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" run sq1 "$@" || exit 1
PMC="SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS" run sq2 "$@" || exit 1
