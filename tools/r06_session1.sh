#!/bin/bash
# round 6, GPU session 1: the suite on the round's first library, the blur-weight pin, and where the 1080p launch's time goes
export TMPDIR=/tmp; root=$(pwd); o=gpurun_out/s1; mkdir -p $o
python -m pytest tests -m gpu -x -q -s -k "hostile" > $o/pytest_hostile.txt 2>&1; tail -15 $o/pytest_hostile.txt
python -m pytest tests -m gpu -x -q > $o/pytest_gpu.txt 2>&1; tail -5 $o/pytest_gpu.txt
python tools/blur_weights_pin.py > $o/blur_weights_pin.txt 2>&1; cat $o/blur_weights_pin.txt
python tools/perf_configs.py 1 2>/dev/null | grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" > $o/cfg1.json
python tools/perf_configs.py 2 2>/dev/null | grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" > $o/cfg2.json
python - <<'PY'
import json
for c in (1,2):
    d=json.load(open(f"gpurun_out/s1/cfg{c}.json")); k=list(d)[0]; print(k, d[k]["frame_us"], d[k]["kernel_us"], d[k]["roofline"]["frac"])
PY
FIGDRAW_HIP_LIB=build/libfigdraw_hip_timing.so python tools/wave_timeline.py 1920 1080 > $o/timeline_1080.txt 2>&1; cat $o/timeline_1080.txt
FIGDRAW_HIP_LIB=build/libfigdraw_hip_timing.so python tools/wave_timeline.py 3840 2160 1 > $o/timeline_4k.txt 2>&1; cat $o/timeline_4k.txt
FIGDRAW_HIP_LIB=build/libfigdraw_hip_stats.so python tools/strip_stats.py bench1080 > $o/strip_stats_1080.txt 2>&1; cat $o/strip_stats_1080.txt
# VALU / SALU wave-instructions of the 1080p launch
out=$root/$o/pmc1080; rm -rf $out; mkdir -p $out
(cd /tmp && W=1920 H=1080 BLUR=0 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_VALU --output-format csv -d $out -o run -- python3 $root/tools/one_frame.py > $out.log 2>&1)
f=$(ls $out/*counter_collection.csv $out/*/*counter_collection.csv 2>/dev/null | head -1)
[ -n "$f" ] && python3 tools/pmc_summary.py $f | grep -A6 "k_composite_tiles<4, true>\|k_bin_draws" | tee $o/pmc1080.txt
rm -rf $out
