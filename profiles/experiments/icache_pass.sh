# Instruction-cache and scalar-cache counters of the headline frame, per-tile kernel and one-block-per-tile kernel:
#   bash profiles/experiments/icache_pass.sh
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
out=gpurun_out/icache
mkdir -p $out
timeout -s KILL 60 rocprofv3 -L > $out/counters.txt 2>&1
grep -io "SQC_[A-Z_]*\|SQ_IFETCH[A-Z_]*\|SQ_INST_LEVEL[A-Z_]*\|SQ_WAIT_IFETCH[A-Z_]*" $out/counters.txt | sort -u > $out/names.txt
cat $out/names.txt
for v in 3 0; do
  RT_RENDER_VARIANT=$v timeout -s KILL 120 rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES --output-format csv -d $out/v$v -o pmc -- \
      python3 bench.py --steps 2 --warmup 3 --no-cpu-baseline > $out/v$v.log 2>&1 || { echo "pass $v failed"; tail -5 $out/v$v.log; exit 1; }
  python3 profiles/summarise_pmc.py $out "k_render" > $out/summary_v$v.json 2>/dev/null
  echo "variant $v ok"
done
