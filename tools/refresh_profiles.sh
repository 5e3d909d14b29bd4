#!/bin/bash
# GPU-box helper: everything profiles/ holds for one tag, in one call.  usage: bash tools/refresh_profiles.sh r01_m
tag=${1:-rXX}; root=$(pwd); out=$root/gpurun_out/$tag; mkdir -p $out; export TMPDIR=/tmp
# counters first: bench.py attaches the newest profiles/r*_pmc_traffic.json and says whether it profiled this very library
bash tools/pmc_traffic.sh $tag > /dev/null 2>&1; cp gpurun_out/${tag}_pmc_traffic.json $out/; cp gpurun_out/${tag}_pmc_traffic.json profiles/
python3 bench.py > $out/bench.log 2>&1; grep '^{"metric"' $out/bench.log | tail -1 > $out/${tag}_bench.json
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof4 -o r -- python3 $root/bench.py --steps 100 --warmup 10 --repeats 2 > $out/bench_rocprof.log 2>&1); grep '^{"metric"' $out/bench_rocprof.log | tail -1 > $out/${tag}_bench_under_rocprof.json
cp $(ls $out/prof4/*kernel_stats.csv $out/prof4/*/*kernel_stats.csv 2>/dev/null | head -1) $out/${tag}_kernel_stats.csv
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof1 -o r -- python3 $root/bench.py --steps 100 --warmup 10 --repeats 2 --frames-in-flight 1 --no-cpu-baseline > /dev/null 2>&1)
cp $(ls $out/prof1/*kernel_stats.csv $out/prof1/*/*kernel_stats.csv 2>/dev/null | head -1) $out/${tag}_kernel_stats_one_frame_at_a_time.csv
# the per-dispatch trace (every launch with its start / end timestamps) of a short one-frame-at-a-time run: small enough to commit
(cd /tmp && rocprofv3 --kernel-trace --output-format csv -d $out/proft -o r -- python3 $root/bench.py --steps 20 --warmup 5 --repeats 1 --frames-in-flight 1 --no-cpu-baseline > /dev/null 2>&1)
cp $(ls $out/proft/*kernel_trace.csv $out/proft/*/*kernel_trace.csv 2>/dev/null | head -1) $out/${tag}_kernel_trace_one_frame_at_a_time.csv
bash tools/run_pmc.sh $tag > $out/${tag}_sq_counters.txt 2>&1
rm -rf $out/prof4 $out/prof1 $out/proft
ls -la $out; cat $out/${tag}_bench.json
