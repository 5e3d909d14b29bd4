#!/bin/bash
# GPU-box helper: bench.py on variant libraries under build/ (make variant NAME=x), same box, interleaved twice.  usage: bash tools/ab_bench_libs.sh x y ..
for rep in 1 2; do for v in "" $*; do
  lib=$PWD/figdraw_amd/libfigdraw_hip.so; [ -n "$v" ] && lib=$PWD/build/libfigdraw_hip_$v.so
  printf "%-8s " "${v:-product}"
  FIGDRAW_HIP_LIB=$lib timeout 300 python3 bench.py --steps 200 --warmup 20 --repeats 5 --no-cpu-baseline 2>&1 | tail -1 | python3 -c "
import sys,json; d=json.loads(sys.stdin.read()); print('value', d['value'], 'replay', d['replay_resident_records']['value'], 'one at a time', d['one_frame_at_a_time']['ms_per_step'], 'H', d['roofline_blur']['passes']['horizontal']['ms'], 'V', d['roofline_blur']['passes']['vertical']['ms'], 'diff', d['frames_in_flight_check']['pixels_differing'])"
done; done
