#!/bin/bash
# GPU-box helper: tools/perf_configs.py for every config (-> gpurun_out/<tag>_configs.json) and one rocprofv3 --kernel-trace --stats
# summary per config (-> gpurun_out/<tag>_config<n>_kernel_stats.csv).  usage: bash tools/refresh_configs.sh r03
tag=${1:-rXX}; root=$(pwd); export TMPDIR=/tmp
python3 tools/perf_configs.py 2>/dev/null | grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" > gpurun_out/${tag}_configs.json
for c in 1 2 3 4 5 6 7 8 9 10 11; do
  out=$root/gpurun_out/cfgprof_$c; rm -rf $out; mkdir -p $out
  (cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $out -o r -- python3 $root/tools/perf_configs.py $c > /dev/null 2>&1)
  cp $(ls $out/*kernel_stats.csv $out/*/*kernel_stats.csv 2>/dev/null | head -1) gpurun_out/${tag}_config${c}_kernel_stats.csv
  rm -rf $out
done
ls -la gpurun_out/${tag}_config*
