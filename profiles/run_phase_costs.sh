#!/bin/bash
# Instruction counters of profiles/experiments/phase_costs.py, one rocprofv3 pass per counter group:
#   bash profiles/run_phase_costs.sh <tag>   ->  gpurun_out/phase_<tag>/{sq1,sq2}/..., table by profiles/phase_table.py
tag=${1:-x}
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/phase_$tag
run() {
    name=$1; shift
    timeout -s KILL 150 rocprofv3 --pmc "$@" --output-format csv -d gpurun_out/phase_$tag/$name -o pmc -- \
        python3 profiles/experiments/phase_costs.py > gpurun_out/phase_$tag/$name.log 2>&1 || { echo "pass $name failed"; tail -5 gpurun_out/phase_$tag/$name.log; return 1; }
    echo "pass $name ok"
}
run sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY || exit 1
run sq2 SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS || exit 1
python3 profiles/phase_table.py gpurun_out/phase_$tag > gpurun_out/phase_$tag/table.txt
cat gpurun_out/phase_$tag/table.txt
