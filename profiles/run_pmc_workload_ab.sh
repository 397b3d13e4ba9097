#!/bin/bash
# Instruction counters of one workload at reduced size for two builds of the library (A/B):
#   bash profiles/run_pmc_workload_ab.sh <tag> <workload> <divisor> <other-lib.so>
tag=$1; wl=$2; div=$3; other=$4
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
for which in new old; do
    if [ $which = old ]; then export RT_AMD_LIB=$GRAFT_REPO_ROOT/$other; else unset RT_AMD_LIB; fi
    out=gpurun_out/ab_${tag}_$which
    mkdir -p $out
    timeout -s KILL 150 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $out/sq1 -o pmc -- python3 profiles/experiments/render_workload.py $wl $div 3 > $out/sq1.log 2>&1 || { echo "$which sq1 failed"; tail -3 $out/sq1.log; exit 1; }
    timeout -s KILL 150 rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_BRANCH --output-format csv -d $out/sq2 -o pmc -- python3 profiles/experiments/render_workload.py $wl $div 3 > $out/sq2.log 2>&1 || { echo "$which sq2 failed"; tail -3 $out/sq2.log; exit 1; }
    grep "frame 2" $out/sq1.log
    python3 profiles/summarise_pmc.py $out "k_render" | python3 -c "
import json,sys
d=json.load(sys.stdin)['counters']; w=d['SQ_WAVES']['per_launch']
print('$which', {k.replace('SQ_',''): round(v['per_launch']/w,1) for k,v in d.items() if k!='SQ_WAVES'}, 'waves', w)"
done
