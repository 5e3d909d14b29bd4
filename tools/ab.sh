#!/bin/bash
# GPU-box helper: A/B the working tree's library against build/libfigdraw_hip_<name>.so on the SAME box, alternating runs.
# usage: bash tools/ab.sh <name> [repeats]
name=${1:-head}; reps=${2:-3}
one() { FIGDRAW_HIP_LIB=$1 python bench.py --steps 100 --warmup 10 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "
import sys,json; d=json.loads(sys.stdin.read()); k=d['frame']['kernel_ms']; print('$2', d['value'], d['one_frame_at_a_time']['value'], k['composite_main'], k['blur_h'], k['blur_v'], k['bin'])"; }
for i in $(seq $reps); do one $(pwd)/build/libfigdraw_hip_$name.so $name; one $(pwd)/figdraw_amd/libfigdraw_hip.so tree; done
