#!/bin/bash
export TMPDIR=/tmp; root=$(pwd); o=gpurun_out/s9; mkdir -p $o
python -m pytest tests -m gpu -x -q > $o/pytest_gpu.txt 2>&1; grep -n "passed\|failed" $o/pytest_gpu.txt | tail -3
python bench.py --steps 20 --warmup 5 > $o/bench.json 2> $o/bench.err; python -c "
import json; d=json.load(open('$o/bench.json')); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['one_frame_at_a_time']['ms_per_step'], d['cpu_baseline']['parity_max_lsb'], d['dtype'], d['roofline_blur'].get('weights_bits'))"
FIGDRAW_HIP_LIB=build/libfigdraw_hip_timing.so python tools/wave_timeline.py 1920 1080 2>&1 | head -8
