#!/bin/bash
# GPU-box helper: blur routes with four contexts in flight -- auto (two-pass routes beside other contexts), always the one-kernel routes, never
for rep in 1 2 3; do for v in auto 1 0; do
  printf "FDH_BLUR_FUSED=%-5s " $v
  if [ $v = auto ]; then unset FDH_BLUR_FUSED; else export FDH_BLUR_FUSED=$v; fi
  timeout 300 python3 bench.py --steps ${STEPS:-200} --warmup ${WARMUP:-20} --repeats 5 --no-cpu-baseline 2>&1 | tail -1 | python3 -c "
import sys,json; d=json.loads(sys.stdin.read()); print('value', d['value'], 'per call', d['per_call_path']['value'], 'replay', d['replay_resident_records']['value'], 'one at a time', d['one_frame_at_a_time']['ms_per_step'], 'diff', d['frames_in_flight_check']['pixels_differing'])"
done; done
