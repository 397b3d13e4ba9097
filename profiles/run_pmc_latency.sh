#!/bin/bash
# Counter passes about what the waves of the headline bench wait for: instruction fetch, scalar data cache, outstanding
# scalar / LDS / vector memory instructions.  One rocprofv3 --pmc pass per group, each a run of its own, never with tracing.
#   bash profiles/run_pmc_latency.sh <tag>      -> gpurun_out/pmc_<tag>/<group>/
tag=${1:-lat}
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/pmc_$tag
run() {
    name=$1; shift
    timeout -s KILL 120 rocprofv3 --pmc "$@" --output-format csv -d gpurun_out/pmc_$tag/$name -o pmc -- \
        python3 bench.py --steps 2 --warmup 3 --no-cpu-baseline > gpurun_out/pmc_$tag/$name.log 2>&1 || { echo "pass $name failed"; tail -5 gpurun_out/pmc_$tag/$name.log; return 1; }
    echo "pass $name ok"
}
run level SQ_IFETCH SQ_IFETCH_LEVEL SQ_INST_LEVEL_SMEM SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_INST_CYCLES_SMEM || exit 1
run icache SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE || exit 1
run dcache SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQC_TC_STALL || exit 1
run base SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_INSTS_SALU || exit 1
