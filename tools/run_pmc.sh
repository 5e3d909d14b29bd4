#!/bin/bash
# Collect SQ counters for the bench workload in separate rocprofv3 --pmc passes (kernel-trace only, as the pool requires).
# usage (on the GPU box): bash tools/run_pmc.sh <tag>   -> gpurun_out/pmc_<tag>_<n>/
tag=${1:-x}
root=$(pwd)
export TMPDIR=/tmp
i=0
for set in \
  "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
  "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_ANY" \
  "VALUBusy SALUBusy" ; do
  i=$((i+1))
  out=$root/gpurun_out/pmc_${tag}_$i
  mkdir -p $out
  (cd /tmp && rocprofv3 --kernel-trace --pmc $set --output-format csv -d $out -o run -- python3 $root/bench.py --steps 8 --warmup 2 --repeats 1 --no-cpu-baseline > $out.log 2>&1)
  f=$(ls $out/*counter_collection.csv $out/*/*counter_collection.csv 2>/dev/null | head -1)
  [ -n "$f" ] && python3 $root/tools/pmc_summary.py $f > $out.txt
done
cat $root/gpurun_out/pmc_${tag}_*.txt
