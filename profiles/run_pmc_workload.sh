#!/bin/bash
# Counter passes over a bench workload at a reduced frame size: bash profiles/run_pmc_workload.sh <tag> <workload> <divisor>
tag=$1; wl=$2; div=$3
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/pmc_$tag
run() {
    name=$1; shift
    timeout -s KILL 150 rocprofv3 --pmc "$@" --output-format csv -d gpurun_out/pmc_$tag/$name -o pmc -- \
        python3 profiles/experiments/render_workload.py $wl $div 3 > gpurun_out/pmc_$tag/$name.log 2>&1 || { echo "pass $name failed"; tail -5 gpurun_out/pmc_$tag/$name.log; return 1; }
    echo "pass $name ok"; tail -1 gpurun_out/pmc_$tag/$name.log
}
run sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY || exit 1
run sq2 SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS || exit 1
run sq3 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM_RD || exit 1
run fetch FETCH_SIZE GRBM_GUI_ACTIVE || exit 1
run write WRITE_SIZE TCC_HIT_sum TCC_MISS_sum || exit 1
