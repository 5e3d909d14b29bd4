#!/bin/bash
# GPU-box helper: bench several build/libfigdraw_hip_<name>.so variants round-robin on the same box.  usage: ab2.sh reps name...
reps=$1; shift
one() { FIGDRAW_HIP_LIB=$(pwd)/build/libfigdraw_hip_$1.so python bench.py --steps 100 --warmup 10 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "
import sys,json; d=json.loads(sys.stdin.read()); k=d['frame']['kernel_ms']; print('$1', d['value'], d['one_frame_at_a_time']['value'], k['composite_main'], k['blur_h'], k['blur_v'], k['bin'])"; }
for i in $(seq $reps); do for n in "$@"; do one $n; done; done
