#!/bin/bash
# HBM traffic per kernel launch of the bench workload, the way MI355X_MICROARCH.md prescribes: separate rocprofv3 --pmc
# passes (kernel-trace only), FETCH_SIZE doubled on gfx950.  Run on the GPU box from the repo root:
#   bash tools/pmc_traffic.sh <tag>     -> gpurun_out/<tag>_pmc_traffic.json  (copy to profiles/ to have bench.py use it)
tag=${1:-rXX}
root=$(pwd)
export TMPDIR=/tmp
for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_VALU_MFMA_BUSY_CYCLES"; do
  name=$(echo $set | cut -d' ' -f1)
  out=$root/gpurun_out/traffic_${tag}_$name
  mkdir -p $out
  (cd /tmp && rocprofv3 --kernel-trace --pmc $set --output-format csv -d $out -o run -- python3 $root/bench.py --steps 8 --warmup 2 --repeats 1 --no-cpu-baseline > $out.log 2>&1)
done
python3 $root/tools/pmc_traffic.py $tag $root/gpurun_out/traffic_${tag}_*/run_counter_collection.csv > $root/gpurun_out/${tag}_pmc_traffic.json
cat $root/gpurun_out/${tag}_pmc_traffic.json
