#!/bin/bash
# Hardware counters of the headline bench, one rocprofv3 --pmc pass per counter group (each pass a
# run of its own; never combined with tracing).  Run on the GPU box from the repo root:
#   bash profiles/run_pmc.sh <tag>
# Writes gpurun_out/pmc_<tag>/<group>/ ; summarise with profiles/summarise_pmc.py.
tag=${1:-r01}
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
run sq3 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM_RD || exit 1
run fetch FETCH_SIZE GRBM_GUI_ACTIVE || exit 1
run write WRITE_SIZE TCC_HIT_sum TCC_MISS_sum || exit 1
