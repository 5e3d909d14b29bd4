bash tools/refresh_profiles.sh r02_g > gpurun_out/refresh.log 2>&1
python3 tools/perf_configs.py > gpurun_out/r02_g/r02_configs.json 2> gpurun_out/r02_g/perf_configs.log
export TMPDIR=/tmp
root=$(pwd); out=$root/gpurun_out/r02_g
for n in 1 2 3 4 5; do
  (cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $out/pc$n -o r -- python3 $root/tools/perf_configs.py $n > /dev/null 2>&1)
  cp $(ls $out/pc$n/*kernel_stats.csv $out/pc$n/*/*kernel_stats.csv 2>/dev/null | head -1) $out/r02_config${n}_kernel_stats.csv
  rm -rf $out/pc$n
done
python3 bench.py --mode stripes --width 7680 --height 4320 --steps 40 --warmup 4 2>&1 | tail -1 > $out/r02_bench_stripes_8k.json
ls $out; tail -1 gpurun_out/refresh.log | cut -c1-300
